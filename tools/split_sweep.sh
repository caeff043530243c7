#!/bin/bash
# lab: candidate-part sizes with the re-score tail on the policy stream (four streams in all) on the pipelined plan step
set -uo pipefail
export M3PC_LIB=$PWD/m3pc_amd/libm3pc_hip_lab.so
for sp in "" "334,334" "342,341" "256,256,256"; do
  for kw in "dict(tail_stream=False)" "dict()"; do
    echo "== split '$sp' $kw"
    M3PC_STREAM_SPLIT=$sp timeout -k 10 120 python tools/pipeline_probe.py 60 2,2 "$kw" 2>&1 | grep depth
  done
done
