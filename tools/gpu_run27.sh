#!/bin/bash
mkdir -p gpurun_out/r3d
cd $GRAFT_REPO_ROOT
export OPENBLAS_NUM_THREADS=1
LRZGPU_TRACE=1 timeout -s ABRT 420 python -X faulthandler -m pytest tests/test_backend_gpu.py tests/test_chunks_gpu.py tests/test_compress_gpu.py "tests/test_roundtrip_gpu.py::test_roundtrip_full_size_headline_workload" -x -q -m gpu > gpurun_out/r3d/a.txt 2> gpurun_out/r3d/a.err
echo "rc=$?" >> gpurun_out/r3d/a.txt
grep "lrzgpu pools" gpurun_out/r3d/a.err | tail -20 > gpurun_out/r3d/pools.txt
grep -v "lrzgpu scan: seg" gpurun_out/r3d/a.err | tail -c 3000 >> gpurun_out/r3d/pools.txt
rm -f gpurun_out/r3d/a.err
