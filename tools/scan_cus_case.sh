#!/bin/bash
# LRZGPU_SCAN_EXCLUSIVE_CUS on the one-chunk configuration (cfg 2: the resolver is the critical path): exclusive first,
# then the default, one warm-up + one timed step each.
mkdir -p gpurun_out/scancus
for n in 4 0; do
  LRZGPU_SCAN_EXCLUSIVE_CUS=$n timeout 40 python bench.py --workload cfg2 --steps 1 --warmup 1 --no-cpu-baseline --no-file-leg > gpurun_out/scancus/cfg2_excl$n.json 2> gpurun_out/scancus/cfg2_excl$n.err
  python - gpurun_out/scancus/cfg2_excl$n.json $n <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    r = d["roofline"]
    print("exclusive CUs %s: %.1f MB/s %.0f ms/step  k_resolve %.0f ms/step (%.1f ms per launch)  k_bt %.0f ms/step" % (
        sys.argv[2], d["value"], d["ms_per_step"], r["per_kernel_ms_per_step"]["k_resolve"], r["avg_launch_ms"], r["per_kernel_ms_per_step"]["k_bt"]))
except Exception as e:
    print("exclusive CUs %s: no line (%s)" % (sys.argv[2], e))
PY
done | tee gpurun_out/scancus/cfg2.log
