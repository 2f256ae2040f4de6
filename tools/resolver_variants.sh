#!/bin/bash
# A/B of resolver builds on one GPU box: tools/_bin/liblrzgpu_<v>.so (built by hand from patched copies of csrc) against
# the tree's own library ("base"), tools/resolve_prof.py on 256 MiB of text and 2 GiB of random bytes, twice each.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/variants
cp lrzip-next_amd/liblrzgpu.so /tmp/liblrzgpu_base.so
for rep in 1 2; do
for v in base "$@"; do
  if [ $v = base ]; then cp /tmp/liblrzgpu_base.so lrzip-next_amd/liblrzgpu.so; else cp tools/_bin/liblrzgpu_$v.so lrzip-next_amd/liblrzgpu.so; fi
  for w in "256 text" "2048 random"; do
    echo -n "$v $w: "; RESOLVE_LAPS=0 python tools/resolve_prof.py $w 2>/dev/null | grep "^scan" | sed 's/.*k_resolve/k_resolve/; s/ in .*//'
  done
done
done 2>&1 | tee gpurun_out/variants/resolver_variants.log
cp /tmp/liblrzgpu_base.so lrzip-next_amd/liblrzgpu.so
