#!/bin/bash
# same-box A/B of lab environment switches (run on the GPU box; needs the lab library): tools/ab_env.sh "" "M3PC_NO_BF16_RESIDUAL=1" ...
# every setting three times, interleaved; prints value, ms_per_step, serial steps/s, p50 latency, alone launch us, re-scored mean, delta
export M3PC_LIB=${M3PC_LIB:-m3pc_amd/libm3pc_hip_lab.so}
for i in 1 2 3; do
for E in "$@"; do
( [ -n "$E" ] && export $E; timeout -k 10 200 python3 bench.py --steps 60 --warmup 8 --no-extras --no-cpu-baseline ${BENCH_FLAGS:-} 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$E]', d['value'], d['ms_per_step'], d['serial_steps_per_s'], d['latency_ms']['p50'], ((d.get('roofline') or {}).get('alone') or {}).get('avg_launch_us'), (d.get('rescore') or {}).get('n_mean'), (d.get('rescore') or {}).get('delta'))" )
done; done
