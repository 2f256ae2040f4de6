#!/bin/bash
mkdir -p gpurun_out/r2z
timeout 2400 python -X faulthandler -m pytest tests -x -v -m gpu > gpurun_out/r2z/full.txt 2>&1
echo "rc=$?" >> gpurun_out/r2z/full.txt
grep -n "PASSED\|FAILED" gpurun_out/r2z/full.txt | tail -3 > gpurun_out/r2z/last.txt
grep -n -m3 -B2 -A30 "Fatal Python\|Memory access fault\|HIP error\|core dumped" gpurun_out/r2z/full.txt | cut -c1-250 >> gpurun_out/r2z/last.txt
dmesg 2>/dev/null | tail -5 >> gpurun_out/r2z/last.txt
