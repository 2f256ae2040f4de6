#!/bin/bash
mkdir -p gpurun_out/r3c
cd $GRAFT_REPO_ROOT
export OPENBLAS_NUM_THREADS=1 OMP_NUM_THREADS=1
timeout 600 /opt/rocm/bin/rocgdb -batch -ex "set pagination off" -ex "handle SIGUSR1 nostop noprint" -ex run -ex "bt 30" -ex "info registers rip" -ex "x/6i \$rip" --args python -m pytest tests/test_backend_gpu.py tests/test_chunks_gpu.py tests/test_compress_gpu.py "tests/test_roundtrip_gpu.py::test_roundtrip_full_size_headline_workload" -x -q -m gpu > gpurun_out/r3c/gdb.txt 2>&1
echo "rc=$?" >> gpurun_out/r3c/gdb.txt
grep -v "^\[New Thread\|^\[Thread .* exited\|^warning: \|^$\|^\[Detaching\|^\[Attaching" gpurun_out/r3c/gdb.txt > gpurun_out/r3c/gdb_f.txt
grep -n -A60 "received signal" gpurun_out/r3c/gdb_f.txt | head -120 > gpurun_out/r3c/gdb_sig.txt
tail -5 gpurun_out/r3c/gdb_f.txt >> gpurun_out/r3c/gdb_sig.txt
rm -f gpurun_out/r3c/gdb.txt gpurun_out/r3c/gdb_f.txt
