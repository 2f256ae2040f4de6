#!/bin/bash
mkdir -p gpurun_out/r2q
exec > gpurun_out/r2q/log.txt 2>&1
set -x
timeout 1500 python -m pytest tests/test_rzip_gpu.py tests/test_chunks_gpu.py tests/test_sweep_gpu.py -x -q -m gpu 2>&1 | tail -4
python tools/resolve_prof.py 64
LRZGPU_RESOLVE_PROF=0 python tools/resolve_prof.py 64 | head -3
LRZGPU_RESOLVE_PROF=0 python tools/resolve_prof.py 512 | head -3
