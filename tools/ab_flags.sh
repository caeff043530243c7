#!/bin/bash
# same-box A/B of bench.py flag sets (run on the GPU box): tools/ab_flags.sh "" "--no-certify-sample" "--race-min 3" ...
# every set three times, interleaved; prints value, ms_per_step, serial steps/s, alone launch us, re-scored mean
for i in 1 2 3; do
for F in "$@"; do
timeout -k 10 200 python3 bench.py --steps 60 --warmup 8 --no-extras --no-cpu-baseline $F 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$F]', d['value'], d['ms_per_step'], d['serial_steps_per_s'], d['latency_ms']['p50'], d['roofline'].get('alone',{}).get('avg_launch_us'), d['rescore']['n_mean'])"
done; done
