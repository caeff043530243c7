timeout 300 python -m pytest tests/test_block_fused_gpu.py -x -q -m gpu 2>&1 | tail -2
M3PC_LIB=$PWD/m3pc_amd/libm3pc_hip_lab.so python tools/block_bench.py 32768 2>&1 | grep -v amdgpu.ids | grep "wave 0\|variant"
B="python bench.py --no-extras --no-cpu-baseline --steps 300 --warmup 30"
run() { echo "$1 $($B 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])')"; }
run a; run b
