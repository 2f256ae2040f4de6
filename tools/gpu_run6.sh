#!/bin/bash
mkdir -p gpurun_out/r2f
exec > gpurun_out/r2f/log.txt 2>&1
set -x
LRZGPU_TRACE=1 timeout 1200 python bench.py --steps 3 --warmup 1 --verify 2> gpurun_out/r2f/bench16g.err | tee gpurun_out/r2f/bench16g.json | cut -c1-600
grep "lrzgpu driver" gpurun_out/r2f/bench16g.err | cut -c1-700
# the N-rank code path of bench.py on this one GPU (gloo hand-off, ranks share the device): functional check only
LRZGPU_BENCH_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 1 --warmup 0 --mib 2048 --window 3 --verify 2> gpurun_out/r2f/bench_n2_gloo.err | tee gpurun_out/r2f/bench_n2_gloo.json | cut -c1-900
tail -3 gpurun_out/r2f/bench_n2_gloo.err
timeout 600 python bench.py --steps 1 --mib 2048 --window 3 --no-cpu-baseline --verify 2>/dev/null | cut -c1-300
