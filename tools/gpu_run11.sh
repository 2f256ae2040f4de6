#!/bin/bash
mkdir -p gpurun_out/r2k
exec > gpurun_out/r2k/log.txt 2>&1
set -x
( time timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 ) 2>&1
bash tools/gpu_run10.sh
cat gpurun_out/r2j/log.txt | grep -v "^+" | cut -c1-420
