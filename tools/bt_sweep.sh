#!/bin/bash
# k_bt_wave cut-over sweep on the headline workload (1 timed step each after 1 warm-up, no CPU leg)
mkdir -p gpurun_out
for m in ${SWEEP:-4096 1024 1000000000}; do
  echo "== LRZGPU_BT_MIN=$m"
  LRZGPU_BT_MIN=$m timeout 400 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-file-leg > gpurun_out/bt_$m.json 2> gpurun_out/bt_$m.err
  python - <<PY
import json
try:
    d=json.load(open("gpurun_out/bt_$m.json"))
    r=d["roofline"]
    print(d["value"], d["ms_per_step"], "k_bt sum", r["per_kernel_ms_per_step"]["k_bt"], "union", r["per_kernel_wall_union_ms_per_step"], "peak", r["per_kernel_peak_concurrency"], "crit", d["critical_path"])
except Exception as e:
    print("failed", e)
PY
done
