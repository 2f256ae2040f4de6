"""Scan-only timing of the resolver with per-phase shader-clock counters (LRZGPU_TRACE=3; RESOLVE_LAPS=0 in the environment of this tool: conflict kinds instead).
usage: python tools/resolve_prof.py [MiB] [text|random]   -- the bench text (or seeded random bytes), one chunk, rzip level 7"""
import ctypes as C
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LAPS = os.environ.get("RESOLVE_LAPS", "1") == "1"
if LAPS:
    os.environ["LRZGPU_TRACE"] = "3"
import torch
import bench

mib = int(sys.argv[1]) if len(sys.argv) > 1 else 64
B = bench.load_bindings()
L = B.lib()
kind = sys.argv[2] if len(sys.argv) > 2 else "text"
if kind == "random":
    import numpy as np
    data = np.random.default_rng(5).integers(0, 256, size=mib << 20, dtype=np.uint8).tobytes()
else:
    data = bench.text_like_torch(mib << 20, 12345, "cuda:0").cpu().numpy().tobytes()
L.lrzgpu_profile_reset()
t0 = time.time()
s0, s1, st, crc, vr = B.hash_search(data, level=7)
dt = time.time() - t0
p = bench.Profile()
L.lrzgpu_profile_get(C.byref(p))
d = [int(v) for v in p.resolve_dbg]
print("scan %d MiB %s: wall %.3f s, k_resolve %.1f ms in %d launches, lookups %d inserts %d" % (mib, kind, dt, p.resolve_ms, p.resolve_launches, p.resolve_lookups, p.resolve_inserts))
names = ("batches", "committed", "serial_steps", "stop_complex", "stop_match", "stop_conflict", "stop_novictim", "stop_sweptrange")
print(dict(zip(names, d[:8])))
if not LAPS:
    print("conflict kinds:", dict(zip(("twin@lane0", "pred not committable", "pred wrote elsewhere", "pred other kind", "same tag lane-1", "same tag earlier", "unrelated write", "over budget/first_conf=0"), d[8:])))
    sys.exit(0)
cyc = d[8:]
tot = sum(cyc) or 1
lab = ("8 refill+publish", "9 top-up+simulate", "10 prefix+sweep", "11 filter reads+suspects", "12 table writes", "13 bookkeeping+serial+shift", "14 stops+filter writes", "15 filter reads")
print("  (mw: lap 7 = suspects publish + barrier: %d ticks)" % d[7])
for n, c in zip(lab, cyc):
    print("  %-28s %12d ticks  %5.1f %%  %8.1f per batch" % (n, c, 100.0 * c / tot, c / max(d[0], 1)))
