#!/bin/bash
# The measurement set of a round (GPU box): PMC + kernel trace of the headline step, the full bench line with the
# whole-file CPU baseline, the file-to-file leg and a timeline, the cfg 2 line, and bench --gpus 2 over gloo on the one GPU.
mkdir -p gpurun_out/final
TAG=${1:-r5}
timeout 1200 python tools/pmc_collect.py $TAG 2>&1 | tail -3
cp gpurun_out/prof/pmc_summary.json profiles/pmc_summary.json   # bench.py reads it from profiles/ (same build id)
timeout 1500 python bench.py --steps 3 --warmup 1 --verify --timeline gpurun_out/final/${TAG}_timeline.csv > gpurun_out/final/${TAG}_bench_n1.json 2> gpurun_out/final/${TAG}_bench_n1.err
tail -c 1500 gpurun_out/final/${TAG}_bench_n1.json; echo
timeout 600 python bench.py --workload cfg2 --steps 2 --warmup 1 --no-cpu-baseline --no-file-leg > gpurun_out/final/${TAG}_bench_cfg2_4g.json 2> gpurun_out/final/${TAG}_bench_cfg2.err
head -c 300 gpurun_out/final/${TAG}_bench_cfg2_4g.json; echo
LRZGPU_BENCH_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 1 --warmup 0 --mib 4096 --base-mib 256 --window 5 --no-cpu-baseline --verify > gpurun_out/final/${TAG}_bench_n2_gloo_functional.json 2> gpurun_out/final/${TAG}_bench_n2.err
head -c 400 gpurun_out/final/${TAG}_bench_n2_gloo_functional.json; echo; tail -3 gpurun_out/final/${TAG}_bench_n2.err
timeout 300 python bench.py --steps 1 --warmup 0 --mib 4096 --base-mib 256 --window 5 --no-cpu-baseline --no-file-leg > gpurun_out/final/${TAG}_bench_n1_same_4g_file.json 2>/dev/null
head -c 300 gpurun_out/final/${TAG}_bench_n1_same_4g_file.json; echo
