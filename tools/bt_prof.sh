#!/bin/bash
# The finder's walk kernels on one 64 MiB block of the bench text under rocprofv3 (GPU box): per-kernel durations of
# k_bt_group<64> / k_bt_group<8> / k_bt for the settings given ("WAVE_MIN,GROUP_MIN" ...), summary under gpurun_out/$1/.
OUT=/root/repo/gpurun_out/${1:-bt}; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for setting in "$@"; do
  tag=$(echo $setting | tr , _)
  rm -rf /tmp/btprof_$tag
  BT_CASE_REPS=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/btprof_$tag -o bt -- python /root/repo/tools/bt_case.py 64 $setting > $OUT/prof_$tag.log 2>&1
  f=$(find /tmp/btprof_$tag -name "*kernel_stats.csv" | head -1)
  echo "== $setting"
  python3 - "$f" <<'PY' | tee $OUT/kernel_stats_$tag.csv
import csv, sys
print("kernel,calls,total_ms,avg_ms,min_ms,max_ms")
for r in csv.reader(open(sys.argv[1])):
    if r[0] != "Name" and ("k_bt" in r[0] or "k_gather" in r[0] or "k_hc5" in r[0]):
        name = r[0].split("(")[0].replace("void ", "").replace("lrzgpu::", "")
        print("%s,%s,%.2f,%.2f,%.2f,%.2f" % (name, r[1], int(r[2]) / 1e6, float(r[3]) / 1e6, int(r[5]) / 1e6, int(r[6]) / 1e6))
PY
done
