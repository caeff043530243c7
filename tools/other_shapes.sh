#!/bin/bash
# the other BASELINE shapes on one GPU (run on the GPU box): T=H=16/8, walker2d critic N=4096, a 2048-candidate halfcheetah shard
set -uo pipefail
B="python3 bench.py --steps 30 --warmup 6 --no-extras --no-cpu-baseline --no-alone-pass"
for d in 2 0; do
  echo "== c2 T=16 H=8 depth $d";  $B --depth $d --traj-length 16 --horizon 8 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
  echo "== c3 walker2d critic N=4096 depth $d"; $B --depth $d --env walker2d --guidance critic_lambda_guiding --candidates 4096 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
  echo "== c4 shard halfcheetah N=2048 H=32 T=64 depth $d"; $B --depth $d --env halfcheetah --candidates 2048 --horizon 32 --traj-length 64 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
done
