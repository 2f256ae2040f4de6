#!/bin/bash
mkdir -p gpurun_out/r2y
exec > gpurun_out/r2y/log.txt 2>&1
for f in test_backend_gpu test_compress_gpu test_roundtrip_gpu test_stream_api_gpu test_configs_gpu; do
echo "=== $f"
timeout 1200 python -m pytest tests/$f.py -x -q -m gpu 2>&1 | grep -v "^  File\|^Extension" | tail -25
done
