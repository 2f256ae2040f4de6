#!/bin/bash
mkdir -p gpurun_out/r2t
exec > gpurun_out/r2t/log.txt 2>&1
set -x
LRZGPU_RESOLVE_PROF=0 timeout 300 python tools/resolve_prof.py 64
timeout 1500 python -m pytest tests/test_rzip_gpu.py tests/test_chunks_gpu.py tests/test_sweep_gpu.py -x -q -m gpu 2>&1 | tail -4
LRZGPU_TRACE=1 timeout 600 python bench.py --workload cfg2 --steps 1 --warmup 0 --no-cpu-baseline 2> gpurun_out/r2t/cfg2.err | cut -c1-130
grep "lrzgpu scan: seg" gpurun_out/r2t/cfg2.err | head -3 | cut -c1-150
