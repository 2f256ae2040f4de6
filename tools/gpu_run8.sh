#!/bin/bash
mkdir -p gpurun_out/r2h
exec > gpurun_out/r2h/log.txt 2>&1
set -x
for v in mask nomask; do
  if [ $v = nomask ]; then export LRZGPU_NO_SCAN_CU_MASK=1; fi
  LRZGPU_TRACE=1 timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --verify 2> gpurun_out/r2h/bench_$v.err | cut -c1-120
  grep "lrzgpu scan: seg \[1,4194305)" gpurun_out/r2h/bench_$v.err | tail -8 | cut -c1-110
  grep "lrzgpu driver" gpurun_out/r2h/bench_$v.err | cut -c1-400
  grep -o '"round_trip_ok": [a-z]*' gpurun_out/r2h/bench_$v.err
done
