#!/bin/bash
# lab: candidate-part sizes x hardware queue count on the pipelined plan step (run on the GPU box)
set -uo pipefail
export M3PC_LIB=$PWD/m3pc_amd/libm3pc_hip_lab.so
for q in 4 8; do
  for sp in "" "334,334" "256,256,256" "668" "334,334,334"; do
    echo "== queues $q split '$sp'"
    GPU_MAX_HW_QUEUES=$q M3PC_STREAM_SPLIT=$sp timeout -k 10 120 python tools/pipeline_probe.py 60 0,2,2 2>&1 | grep depth
  done
done
