#!/bin/bash
mkdir -p gpurun_out/r3a
cd $GRAFT_REPO_ROOT
LRZGPU_TRACE=2 timeout -s ABRT 420 python -X faulthandler -m pytest tests/test_configs_gpu.py "tests/test_roundtrip_gpu.py::test_roundtrip_full_size_headline_workload" -x -v -m gpu > gpurun_out/r3a/a.txt 2> gpurun_out/r3a/a.err
echo "rc=$?" >> gpurun_out/r3a/a.txt
tail -c 6000 gpurun_out/r3a/a.err > gpurun_out/r3a/a_err_tail.txt
grep -c "" gpurun_out/r3a/a.err >> gpurun_out/r3a/a.txt
rm -f gpurun_out/r3a/a.err
