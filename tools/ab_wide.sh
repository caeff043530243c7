#!/bin/bash
# same-box A/B of the T = 64 pipelined attention (attn_bf16_pipe_wide_kernel) on a config-4 shard and the full config 4: lab build, switch M3PC_NO_ATTN_PIPE_WIDE
mkdir -p gpurun_out
export M3PC_LIB=m3pc_amd/libm3pc_hip_lab.so
for r in 1 2; do
  echo "== wide"; python tools/ab_legs.py c4_shard c4_full
  echo "== direct"; M3PC_NO_ATTN_PIPE_WIDE=1 python tools/ab_legs.py c4_shard c4_full
done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_ab_attn_wide.txt
