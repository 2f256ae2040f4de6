#!/bin/bash
# look-ahead wave: parity + dense-phase timing with and without it (cfg2 = one chunk alone; cfg3 = eight side by side)
mkdir -p gpurun_out/r2o
exec > gpurun_out/r2o/log.txt 2>&1
set -x
timeout 1200 python -m pytest tests/test_rzip_gpu.py -x -q -m gpu 2>&1 | tail -4
for la in 0 1; do
LRZGPU_NO_PRESIM=$la LRZGPU_TRACE=1 timeout 600 python bench.py --workload cfg2 --steps 1 --warmup 0 --no-cpu-baseline 2> gpurun_out/r2o/cfg2_nola$la.err | cut -c1-130
grep "lrzgpu scan: seg" gpurun_out/r2o/cfg2_nola$la.err | head -6 | cut -c1-150
grep -o '"k_resolve": [0-9.]*' gpurun_out/r2o/cfg2_nola$la.err | head -2
done
for la in 0 1; do
LRZGPU_NO_PRESIM=$la LRZGPU_TRACE=1 timeout 900 python bench.py --steps 1 --warmup 1 --no-cpu-baseline 2> gpurun_out/r2o/cfg3_nola$la.err | cut -c1-130
grep "lrzgpu scan: seg \[1,4194305)" gpurun_out/r2o/cfg3_nola$la.err | tail -8 | cut -c1-110
grep "lrzgpu driver" gpurun_out/r2o/cfg3_nola$la.err | cut -c1-330
done
