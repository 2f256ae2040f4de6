#!/bin/bash
mkdir -p gpurun_out/r2x
exec > gpurun_out/r2x/log.txt 2>&1
set -x
LRZGPU_TRACE=1 timeout 600 python bench.py --workload cfg2 --steps 2 --warmup 1 --no-cpu-baseline --verify 2> gpurun_out/r2x/cfg2.err > gpurun_out/r2x/cfg2.json
cut -c1-200 gpurun_out/r2x/cfg2.json
grep -o '"per_kernel_ms_per_step": {[^}]*}' gpurun_out/r2x/cfg2.json
LRZGPU_TRACE=1 timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --verify 2> gpurun_out/r2x/cfg3.err > gpurun_out/r2x/cfg3.json
cut -c1-200 gpurun_out/r2x/cfg3.json
grep -o '"per_kernel_ms_per_step": {[^}]*}' gpurun_out/r2x/cfg3.json
grep "lrzgpu scan: seg \[1,4194305)" gpurun_out/r2x/cfg3.err | tail -8 | cut -c1-110
grep "lrzgpu driver" gpurun_out/r2x/cfg3.err | cut -c1-330
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -5
