#!/bin/bash
# The certificate sweeps run long, once per round for the record (on the GPU box): tools/long_sweeps.sh
# -> gpurun_out/r06_certificate_sweep*_long_*.json (copied to profiles/)
mkdir -p gpurun_out
M3PC_SWEEP_SCALE=6 python -m pytest tests/test_certificate_gpu.py -m gpu -x -q -k "sweep_argmax" > gpurun_out/r06_sweep_long.log 2>&1; echo "long sweep rc $?"; tail -3 gpurun_out/r06_sweep_long.log
M3PC_SWEEP_SCALE=13 python -m pytest tests/test_certificate_gpu.py -m gpu -x -q -k "trained_like" > gpurun_out/r06_sweep_trained_long.log 2>&1; echo "trained-like long sweep rc $?"; tail -3 gpurun_out/r06_sweep_trained_long.log
