#!/bin/bash
mkdir -p gpurun_out/r2l
exec > gpurun_out/r2l/log.txt 2>&1
set -x
( time timeout 2400 python -m pytest tests/test_roundtrip_gpu.py tests/test_rzip_gpu.py tests/test_stream_api_gpu.py tests/test_sweep_gpu.py tests/test_backend_gpu.py -x -q -m gpu 2>&1 | tail -6 ) 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 2400 python $GRAFT_REPO_ROOT/tools/pmc_collect.py r2 2>&1 | tail -4
cd $GRAFT_REPO_ROOT
LRZGPU_TRACE=1 timeout 1200 python bench.py --steps 3 --warmup 1 --verify 2> gpurun_out/r2l/bench16g.err > gpurun_out/r2l/bench16g.json
cut -c1-200 gpurun_out/r2l/bench16g.json
grep "lrzgpu driver" gpurun_out/r2l/bench16g.err | cut -c1-300
python __graft_entry__.py --smoke 2>&1 | tail -2
