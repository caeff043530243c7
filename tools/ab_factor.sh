#!/bin/bash
# same-box price of the calibration factor on the bench workload: tools/ab_factor.sh 1.6 2.1
for i in 1 2 3; do for F in "$@"; do
timeout -k 10 200 python3 bench.py --steps 60 --warmup 8 --no-extras --no-cpu-baseline --calibration-factor $F 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[factor $F]', d['value'], d['ms_per_step'], d['serial_steps_per_s'], d['latency_ms']['p50'], d['rescore']['n_mean'], d['rescore']['n_max'], round(d['rescore']['delta'],3))"
done; done
