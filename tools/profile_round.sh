#!/bin/bash
# The four rocprofv3 passes whose summaries are committed under profiles/ (run on the GPU box through gpurun):
#   kernel trace + stats, FETCH_SIZE, WRITE_SIZE (separate passes: TCC slots), SQ counters.  usage: tools/profile_round.sh r03
set -euo pipefail
: "${GRAFT_REPO_ROOT:?run on the GPU box (gpurun exports GRAFT_REPO_ROOT)}"
TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
B="python3 bench.py --steps 10 --warmup 3 --no-extras --no-cpu-baseline --no-alone-pass"
rocprofv3 --kernel-trace --stats -d gpurun_out/${TAG}_trace -o kt --output-format csv -- $B > gpurun_out/${TAG}_trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d gpurun_out/${TAG}_fetch -o pmc --output-format csv -- $B > gpurun_out/${TAG}_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d gpurun_out/${TAG}_write -o pmc --output-format csv -- $B > gpurun_out/${TAG}_write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_BF16 \
    -d gpurun_out/${TAG}_sq -o pmc --output-format csv -- $B > gpurun_out/${TAG}_sq.log 2>&1
tail -1 gpurun_out/${TAG}_trace.log | cut -c1-200
# one step in the serial order (every kernel of a step in sequence) and a window of the pipelined run (streams side by side)
rocprofv3 --kernel-trace -d gpurun_out/${TAG}_trace_serial -o kt --output-format csv -- $B --depth 0 > gpurun_out/${TAG}_trace_serial.log 2>&1
python3 tools/step_timeline.py gpurun_out/${TAG}_trace_serial gpurun_out/${TAG}_step_timeline.txt > /dev/null
python3 tools/timeline_window.py gpurun_out/${TAG}_trace gpurun_out/${TAG}_pipeline_timeline.txt 2
python3 bench.py --steps 50 --warmup 10 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
tail -c 600 gpurun_out/${TAG}_bench.json
