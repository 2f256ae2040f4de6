#!/bin/bash
# HBM traffic of the finder's walk kernel on one 64 MiB block of the bench text alone on the GPU: rocprofv3 --pmc FETCH_SIZE /
# WRITE_SIZE (separate passes, kernel trace only beside them), per kernel, under gpurun_out/$1/.
OUT=/root/repo/gpurun_out/${1:-btpmc}; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/btpmc_$c
  BT_CASE_REPS=1 timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/btpmc_$c -o bt -- python /root/repo/tools/bt_case.py 64 ${2:-4096,512} > $OUT/pmc_$c.log 2>&1
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
tot = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob("/tmp/btpmc_%s/**/*counter_collection.csv" % c, recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == c:
                k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("lrzgpu::", "")
                if k.startswith("rocprim") or "rocprim" in k: k = "rocprim sorts / scans"
                tot[k][c] += float(r["Counter_Value"])
                if c == "FETCH_SIZE": n[k] += 1
with open(sys.argv[1] + "/bt_block_pmc_hbm.csv", "w") as o:
    o.write("kernel,launches,FETCH_SIZE_MB,WRITE_SIZE_MB\n")
    for k in sorted(tot, key=lambda k: -(tot[k]["FETCH_SIZE"] + tot[k]["WRITE_SIZE"])):
        o.write("%s,%d,%.1f,%.1f\n" % (k, n[k], tot[k]["FETCH_SIZE"] / 1024, tot[k]["WRITE_SIZE"] / 1024))
print(open(sys.argv[1] + "/bt_block_pmc_hbm.csv").read())
PY
