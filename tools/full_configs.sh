#!/bin/bash
# full-size BASELINE configurations (property tests) + the PCIe-inclusive rate; log kept under profiles/
mkdir -p gpurun_out/full
exec > gpurun_out/full/log.txt 2>&1
set -x
rm -f gpurun_out/full/times.txt
LRZGPU_TIMES_FILE=gpurun_out/full/times.txt LRZGPU_FULL_CONFIGS=1 timeout 2400 python -m pytest tests/test_configs_gpu.py -x -q -m gpu -k "full" 2>&1 | tail -8
timeout 900 python tools/host_input_case.py 16384 2>&1 | tail -4
cat gpurun_out/full/times.txt
