#!/bin/bash
# one serial-order kernel trace of the bench step (run on the GPU box): tools/quick_trace.sh <tag> [bench args]
set -euo pipefail
: "${GRAFT_REPO_ROOT:?run on the GPU box}"
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
rocprofv3 --kernel-trace --stats -d gpurun_out/${TAG}_trace -o kt --output-format csv -- python3 bench.py --steps 10 --warmup 3 --no-extras --no-cpu-baseline --no-alone-pass --depth 0 "$@" > gpurun_out/${TAG}_trace.log 2>&1
python3 tools/step_timeline.py gpurun_out/${TAG}_trace gpurun_out/${TAG}_step_timeline.txt > /dev/null
find gpurun_out/${TAG}_trace -name "*kernel_stats.csv" -exec cp {} gpurun_out/${TAG}_kernel_stats.csv \;
rm -rf gpurun_out/${TAG}_trace
