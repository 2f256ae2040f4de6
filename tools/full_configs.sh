#!/bin/bash
# The two full-size property runs that are too long for the GPU suite's budget (run on the GPU box; log for profiles/):
# cfg 4 at 10 GiB (--zstd --zstd-level 15 -w 26, round trip through the library decoder) and the full decode of cfg 5's
# 32 GiB image (every chunk CRC and the MD5 over all of it).
mkdir -p gpurun_out/full
python - <<'PY' 2>&1 | tee gpurun_out/full/times.txt
import hashlib, os, sys, time
sys.path.insert(0, "tests")
import torch, datagen
from conftest import load_bindings
B = load_bindings()
ram = os.sysconf("SC_PHYS_PAGES") * os.sysconf("SC_PAGE_SIZE")
ncpu = os.cpu_count()
data = datagen.source_tree_tar(40, 256 << 20, seed=7)
t0 = time.time()
got, ctl = B.compress_buffer(data, level=7, threads=ncpu, processors=ncpu, ramsize=ram, window=26, zstd=True, zstd_level=15)
dt = time.time() - t0
print("cfg4: %d MiB tar of 40 copies of a 256 MiB synthetic source tree, --zstd --zstd-level 15 -w 26 (rzip level 6), host input: %.1f s = %.1f MB/s, image %d bytes"
      % (len(data) >> 20, dt, (len(data) >> 20) / dt, len(got)), flush=True)
assert B.decompress_buffer(got) == data
print("cfg4: round trip ok", flush=True)
del data, got
n = 32 << 30
g = torch.Generator(device="cuda"); g.manual_seed(5)
buf = torch.empty(n + 256, dtype=torch.uint8, device="cuda")
for o in range(0, n, 1 << 30):
    buf[o:o + (1 << 30)] = torch.randint(-(1 << 63), (1 << 63) - 1, ((1 << 30) // 8,), generator=g, device="cuda", dtype=torch.int64).view(torch.uint8)
torch.cuda.synchronize()
t0 = time.time()
out, ctl = B.compress_device(buf.data_ptr(), n, level=7, threads=ncpu, processors=ncpu, ramsize=ram, window=328, copy=False)
dt = time.time() - t0
print("cfg5: 32 GiB random, -L7 -w 328, one chunk, input in HBM: %.1f s = %.1f MB/s, image %d bytes" % (dt, (n >> 20) / dt, len(out)), flush=True)
t0 = time.time()
back = B.decompress_buffer(out)
assert len(back) == n and hashlib.md5(back).digest() == bytes(ctl.hash_resblock)
print("cfg5: full decode + MD5 ok (%.1f s)" % (time.time() - t0), flush=True)
PY
