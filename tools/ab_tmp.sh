#!/bin/bash
python -m pytest tests -m gpu -x -q 2>&1 | tail -5
export M3PC_LIB=$PWD/m3pc_amd/libm3pc_hip_lab.so
for i in 1 2 3; do
for V in fused unfused; do
if [ $V = unfused ]; then export M3PC_NO_HEAD_F32_FUSED=1; else unset M3PC_NO_HEAD_F32_FUSED; fi
timeout -k 10 200 python3 bench.py --steps 60 --warmup 8 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$V]', d['value'], d['ms_per_step'], d['serial_steps_per_s'], d['latency_ms']['p50'], d['closed_loop']['ms']['p50'], d['latency_ms_shipped']['bf16']['p50'], d['roofline'].get('alone',{}).get('avg_launch_us'))"
done; done
