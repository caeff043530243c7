#!/bin/bash
# The rocprofv3 passes whose summaries are committed under profiles/ (run on the GPU box through gpurun):
#   kernel trace + stats (as the timed region runs), FETCH_SIZE, WRITE_SIZE (separate passes: TCC slots), SQ counters,
#   the serial-order kernel trace (every launch alone: the `alone` durations) and the config-5 leg.  usage: tools/profile_round.sh r05
set -euo pipefail
: "${GRAFT_REPO_ROOT:?run on the GPU box (gpurun exports GRAFT_REPO_ROOT)}"
TAG=${1:-r06}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
B="python3 bench.py --steps 10 --warmup 3 --no-extras --no-cpu-baseline --no-alone-pass"
rocprofv3 --kernel-trace --stats -d gpurun_out/${TAG}_trace -o kt --output-format csv -- $B > gpurun_out/${TAG}_trace.log 2>&1
echo "trace done"
rocprofv3 --pmc FETCH_SIZE -d gpurun_out/${TAG}_fetch -o pmc --output-format csv -- $B > gpurun_out/${TAG}_fetch.log 2>&1
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE -d gpurun_out/${TAG}_write -o pmc --output-format csv -- $B > gpurun_out/${TAG}_write.log 2>&1
echo "write done"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_BF16 \
    -d gpurun_out/${TAG}_sq -o pmc --output-format csv -- $B > gpurun_out/${TAG}_sq.log 2>&1
echo "sq done"
# one step in the serial order (every kernel of a step in sequence, each launch alone) and a window of the pipelined run
rocprofv3 --kernel-trace --stats -d gpurun_out/${TAG}_trace_serial -o kt --output-format csv -- $B --depth 0 > gpurun_out/${TAG}_trace_serial.log 2>&1
python3 tools/step_timeline.py gpurun_out/${TAG}_trace_serial gpurun_out/${TAG}_step_timeline.txt > /dev/null
python3 tools/timeline_window.py gpurun_out/${TAG}_trace gpurun_out/${TAG}_pipeline_timeline.txt 2
echo "serial done"
# every launch ALONE on the chip: serial order AND the two candidate halves one after the other on one stream -- the arrangement of
# bench.py's roofline.frac (6 x the fused tail's average here must fit inside ms_per_step)
rocprofv3 --kernel-trace --stats -d gpurun_out/${TAG}_trace_alone -o kt --output-format csv -- $B --depth 0 --serial-halves > gpurun_out/${TAG}_trace_alone.log 2>&1
echo "alone done"
# BASELINE config 5: 8192 zero-shot windows per call
rocprofv3 --kernel-trace --stats -d gpurun_out/${TAG}_trace_c5 -o kt --output-format csv -- python3 bench.py --config c5 --steps 10 --warmup 2 --no-extras > gpurun_out/${TAG}_trace_c5.log 2>&1
echo "c5 done"
# a 2048-candidate shard of BASELINE config 4 (T = 64: the 97-row attentions), serial order
rocprofv3 --kernel-trace --stats -d gpurun_out/${TAG}_trace_c4 -o kt --output-format csv -- $B --depth 0 --env halfcheetah --candidates 2048 --horizon 32 --traj-length 64 > gpurun_out/${TAG}_trace_c4.log 2>&1
echo "c4 shard done"
python3 bench.py --steps 50 --warmup 10 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
tail -c 400 gpurun_out/${TAG}_bench.json
