#!/bin/bash
mkdir -p gpurun_out/r2s
exec > gpurun_out/r2s/log.txt 2>&1
set -x
timeout 300 python tools/resolve_prof.py 64
LRZGPU_RESOLVE_PROF=0 timeout 300 python tools/resolve_prof.py 64
LRZGPU_NO_PRESIM=1 LRZGPU_RESOLVE_PROF=0 timeout 300 python tools/resolve_prof.py 64 | head -3
timeout 1500 python -m pytest tests/test_rzip_gpu.py tests/test_chunks_gpu.py -x -q -m gpu 2>&1 | tail -4
