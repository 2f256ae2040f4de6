#!/bin/bash
mkdir -p gpurun_out/r2u
exec > gpurun_out/r2u/log.txt 2>&1
set -x
for nw in 4 2 1; do
LRZGPU_RESOLVE_WAVES=$nw LRZGPU_RESOLVE_PROF=0 timeout 120 python tools/resolve_prof.py 64 | head -3
done
LRZGPU_RESOLVE_WAVES=4 timeout 1200 python -m pytest tests/test_rzip_gpu.py -x -q -m gpu 2>&1 | tail -6
LRZGPU_RESOLVE_WAVES=2 timeout 1200 python -m pytest tests/test_rzip_gpu.py -x -q -m gpu 2>&1 | tail -6
