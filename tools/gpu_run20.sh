#!/bin/bash
mkdir -p gpurun_out/r2v
exec > gpurun_out/r2v/log.txt 2>&1
set -x
LRZGPU_RESOLVE_PROF=0 timeout 120 python tools/resolve_prof.py 64 | head -3
LRZGPU_RESOLVE_PROF=0 timeout 200 python tools/resolve_prof.py 1024 | head -3
LRZGPU_RESOLVE_WAVES=1 LRZGPU_RESOLVE_PROF=0 timeout 200 python tools/resolve_prof.py 1024 | head -3
timeout 1500 python -m pytest tests/test_rzip_gpu.py tests/test_chunks_gpu.py tests/test_sweep_gpu.py -x -q -m gpu 2>&1 | tail -4
