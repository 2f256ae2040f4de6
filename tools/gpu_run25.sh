#!/bin/bash
mkdir -p gpurun_out/r3b
cd $GRAFT_REPO_ROOT
for nw in 4 1; do
LRZGPU_RESOLVE_WAVES=$nw LRZGPU_TRACE=1 timeout -s ABRT 400 python -X faulthandler -m pytest tests/test_backend_gpu.py tests/test_chunks_gpu.py tests/test_compress_gpu.py "tests/test_roundtrip_gpu.py::test_roundtrip_full_size_headline_workload" -x -v -m gpu > gpurun_out/r3b/nw$nw.txt 2> gpurun_out/r3b/nw$nw.err
echo "rc=$?" >> gpurun_out/r3b/nw$nw.txt
grep -v "lrzgpu scan: seg" gpurun_out/r3b/nw$nw.err | tail -c 5000 > gpurun_out/r3b/nw${nw}_err_tail.txt
grep "lrzgpu scan: seg" gpurun_out/r3b/nw$nw.err | tail -3 >> gpurun_out/r3b/nw${nw}_err_tail.txt
rm -f gpurun_out/r3b/nw$nw.err
done
