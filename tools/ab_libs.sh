#!/bin/bash
# same-box A/B of builds of the library (run on the GPU box): tools/ab_libs.sh libA.so libB.so ...   (default: _prev against the product)
LIBS=${@:-"libm3pc_hip_prev.so libm3pc_hip.so"}
for i in 1 2 3; do
for L in $LIBS; do
M3PC_LIB=$PWD/m3pc_amd/$L timeout -k 10 200 python3 bench.py --steps 40 --warmup 8 --no-extras --no-cpu-baseline 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$L', d['value'], d['ms_per_step'], d['roofline'].get('alone',{}).get('avg_launch_us'))"
done; done
