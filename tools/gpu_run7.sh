#!/bin/bash
mkdir -p gpurun_out/r2g
exec > gpurun_out/r2g/log.txt 2>&1
set -x
timeout 1500 python -m pytest tests/test_backend_gpu.py tests/test_compress_gpu.py tests/test_sweep_gpu.py -x -q -m gpu 2>&1 | tail -8
LRZGPU_TRACE=1 timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2> gpurun_out/r2g/bench16g.err | tee gpurun_out/r2g/bench16g.json | cut -c1-300
grep "lrzgpu driver" gpurun_out/r2g/bench16g.err | cut -c1-700
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r2g/bench16g.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['per_kernel_ms_per_step'])
PY
