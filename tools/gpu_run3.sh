#!/bin/bash
mkdir -p gpurun_out/r2c
exec > gpurun_out/r2c/log.txt 2>&1
set -x
cat /sys/kernel/mm/transparent_hugepage/enabled /sys/kernel/mm/transparent_hugepage/defrag
timeout 600 python tools/parser_gpu_case.py 64 1,16 2
HUGE=1 timeout 600 python tools/parser_gpu_case.py 64 1,16 2
grep AnonHuge /proc/meminfo
LRZGPU_TRACE=1 timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2> gpurun_out/r2c/bench16g.err | tee gpurun_out/r2c/bench16g.json
grep "lrzgpu driver" gpurun_out/r2c/bench16g.err
LRZGPU_TRACE=1 timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --host-threads 14 2> gpurun_out/r2c/bench16g_14.err | tee gpurun_out/r2c/bench16g_14.json
grep "lrzgpu driver" gpurun_out/r2c/bench16g_14.err
