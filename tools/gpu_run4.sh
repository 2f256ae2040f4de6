#!/bin/bash
mkdir -p gpurun_out/r2d
exec > gpurun_out/r2d/log.txt 2>&1
set -x
timeout 1500 python -m pytest tests/test_chunks_gpu.py tests/test_stream_api_gpu.py tests/test_backend_gpu.py -x -q -m gpu 2>&1 | tail -15
LRZGPU_TRACE=2 timeout 900 python bench.py --steps 1 --warmup 1 --no-cpu-baseline 2> gpurun_out/r2d/bench16g.err | cut -c1-400
python tools/timeline.py gpurun_out/r2d/bench16g.err | tail -60
grep "lrzgpu driver" gpurun_out/r2d/bench16g.err | cut -c1-600
