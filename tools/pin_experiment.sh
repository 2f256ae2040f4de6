#!/bin/bash
# Host-side placement experiment on the GPU box (16-CPU CFS quota over 256 hardware threads):
# default scheduler placement vs one encoder per L3 domain, and a mildly oversubscribed pool.
mkdir -p gpurun_out
run() { name=$1; shift; echo "== $name"; env "$@" python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/pin_$name.json 2> gpurun_out/pin_$name.err; tail -c 600 gpurun_out/pin_$name.json; echo; }
run default X=1
run pinned LRZGPU_PIN_ENCODERS=1
