#!/bin/bash
mkdir -p gpurun_out/r2j
exec > gpurun_out/r2j/log.txt 2>&1
set -x
for ss in 4 5 6; do
  LRZGPU_TRACE=1 timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --scan-slots $ss 2> gpurun_out/r2j/bench_$ss.err | cut -c1-130
  grep "lrzgpu driver" gpurun_out/r2j/bench_$ss.err | cut -c1-420
done
for pin in 1; do
  LRZGPU_PIN_ENCODERS=$pin LRZGPU_TRACE=1 timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2> gpurun_out/r2j/bench_pin$pin.err | cut -c1-130
  grep "lrzgpu driver" gpurun_out/r2j/bench_pin$pin.err | cut -c1-420
done
