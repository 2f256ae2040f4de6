#!/bin/bash
# first GPU pass of round 2: environment probe, new multi-chunk tests, 16 GiB bench
mkdir -p gpurun_out/r2a
exec > gpurun_out/r2a/log.txt 2>&1
set -x
nproc; cat /sys/fs/cgroup/cpu.max; cat /sys/fs/cgroup/memory.max; free -g | head -3; ulimit -l; ulimit -n
rocm-smi --showmeminfo vram | head -8
timeout 1500 python -m pytest tests/test_chunks_gpu.py tests/test_compress_gpu.py -x -q -m gpu 2>&1 | tail -15
LRZGPU_TRACE=1 timeout 600 python bench.py --mib 2048 --steps 1 --no-cpu-baseline --verify 2> gpurun_out/r2a/bench2g.err | tee gpurun_out/r2a/bench2g.json
grep "lrzgpu driver" gpurun_out/r2a/bench2g.err
LRZGPU_TRACE=1 timeout 900 python bench.py --steps 2 --warmup 1 --verify 2> gpurun_out/r2a/bench16g.err | tee gpurun_out/r2a/bench16g.json
grep "lrzgpu driver" gpurun_out/r2a/bench16g.err
