#!/bin/bash
mkdir -p gpurun_out/r2b
exec > gpurun_out/r2b/log.txt 2>&1
set -x
lscpu | head -30
timeout 900 python tools/parser_gpu_case.py 16,64 1,8,16 0,2
