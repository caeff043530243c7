#!/bin/bash
# three times the certificate sweeps of record (tools/long_sweeps.sh), for the statistics: gpurun_out/r06_certificate_sweep*_long_xl_*.json
mkdir -p gpurun_out
export M3PC_SWEEP_TAG=_xl
M3PC_SWEEP_SCALE=18 python -m pytest tests/test_certificate_gpu.py -m gpu -q -k "sweep_argmax" > gpurun_out/r06_sweep_xl.log 2>&1; echo "xl sweep rc $?"; tail -2 gpurun_out/r06_sweep_xl.log
M3PC_SWEEP_SCALE=39 python -m pytest tests/test_certificate_gpu.py -m gpu -q -k "trained_like" > gpurun_out/r06_sweep_trained_xl.log 2>&1; echo "trained-like xl sweep rc $?"; tail -2 gpurun_out/r06_sweep_trained_xl.log
