#!/bin/bash
# The wave-per-bucket launches beside k_bt on a second stream (LRZGPU_BT_OVERLAP=1) against one stream, per cut-over,
# on one 64 MiB block of the bench text; then the finder parity tests with the overlap on.
mkdir -p gpurun_out/btovl
{
echo "== one stream";  timeout 100 python tools/bt_case.py 64 4096 2048 1536 1024 2>&1 | grep -E "wave_min|equal"
echo "== two streams"; LRZGPU_BT_OVERLAP=1 timeout 100 python tools/bt_case.py 64 4096 2048 1536 1024 2>&1 | grep -E "wave_min|equal"
echo "== parity, two streams"
LRZGPU_BT_OVERLAP=1 timeout 150 python -m pytest tests/test_backend_gpu.py -x -q -p no:cacheprovider -k "forced_bucket or lds_resident or match_lists" 2>&1 | tail -3
} | tee gpurun_out/btovl/overlap.log
