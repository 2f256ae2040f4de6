#!/bin/bash
mkdir -p gpurun_out/r2m
exec > gpurun_out/r2m/log.txt 2>&1
set -x
timeout 1500 python -m pytest tests/test_rzip_gpu.py tests/test_chunks_gpu.py tests/test_sweep_gpu.py -x -q -m gpu 2>&1 | tail -4
LRZGPU_TRACE=1 timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --verify 2> gpurun_out/r2m/bench.err | cut -c1-130
grep "lrzgpu scan: seg \[1,4194305)" gpurun_out/r2m/bench.err | tail -8 | cut -c1-110
grep "lrzgpu driver" gpurun_out/r2m/bench.err | cut -c1-330
LRZGPU_TRACE=1 timeout 600 python bench.py --steps 1 --mib 2048 --no-cpu-baseline 2> gpurun_out/r2m/bench2g.err | cut -c1-130
grep "lrzgpu scan: seg" gpurun_out/r2m/bench2g.err | head -4 | cut -c1-110
