#!/bin/bash
# A/B of finder builds on one GPU box: tools/_bin/liblrzgpu_<v>.so against the tree's own library ("base"),
# tools/bt_case.py on one 64 MiB block of the bench text (default cut-overs), twice each.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/variants
cp lrzip-next_amd/liblrzgpu.so /tmp/liblrzgpu_base.so
for rep in 1 2; do
for v in base "$@"; do
  if [ $v = base ]; then cp /tmp/liblrzgpu_base.so lrzip-next_amd/liblrzgpu.so; else cp tools/_bin/liblrzgpu_$v.so lrzip-next_amd/liblrzgpu.so; fi
  echo -n "$v: "; BT_CASE_REPS=2 python tools/bt_case.py 64 4096,512 2>/dev/null | grep "k_bt" | sed 's/ wall.*//'
done
done 2>&1 | tee gpurun_out/variants/bt_variants.log
cp /tmp/liblrzgpu_base.so lrzip-next_amd/liblrzgpu.so
