#!/bin/bash
# k_bt (lane per bucket) with fewer wavefronts per CU (LRZGPU_BT_PAD_LDS: unused dynamic LDS per wave): time on one
# 64 MiB block of the bench text, and the HBM counters for the default and one padded setting.
mkdir -p gpurun_out/btocc
O=gpurun_out/btocc
for pad in 0 8192 21504 30720; do
  echo "== pad $pad"; LRZGPU_BT_PAD_LDS=$pad timeout 100 python tools/bt_case.py 64 4096 2>&1 | grep wave_min
done | tee $O/sweep.log
cd /tmp; export TMPDIR=/tmp
for pad in 0 21504; do for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/occ_$pad_$c; LRZGPU_BT_PAD_LDS=$pad BT_CASE_REPS=1 timeout 120 rocprofv3 --kernel-trace --pmc $c -d /tmp/occ_${pad}_$c -o x --output-format csv -- python $GRAFT_REPO_ROOT/tools/bt_case.py 64 4096 > /dev/null 2>&1
  f=$(find /tmp/occ_${pad}_$c -name "*counter_collection.csv" | head -1)
  python - "$f" $c $pad <<'PY'
import csv, sys
t = 0.0
for r in csv.DictReader(open(sys.argv[1])):
    if r.get("Counter_Name") == sys.argv[2] and "k_bt" in r.get("Kernel_Name", "") and "k_bt_wave" not in r.get("Kernel_Name", ""):
        t += float(r["Counter_Value"])
print("pad %s k_bt %s %.1f KB" % (sys.argv[3], sys.argv[2], t))
PY
done; done | tee -a $GRAFT_REPO_ROOT/$O/sweep.log
