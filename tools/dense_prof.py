"""Where a round of the dense resolver variant spends its shader-clock ticks (LRZGPU_TRACE=3 laps of k_resolve_mw<1, .., true>).
usage: python tools/dense_prof.py [few|phrases|text|...] [MiB] [level]"""
import ctypes as C
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["LRZGPU_TRACE"] = "3"
os.environ.setdefault("LRZGPU_RESOLVE_DENSE", "always")
import bench
import datagen
kind = sys.argv[1] if len(sys.argv) > 1 else "few"
mib = int(sys.argv[2]) if len(sys.argv) > 2 else 5
level = int(sys.argv[3]) if len(sys.argv) > 3 else 7
B = bench.load_bindings()
L = B.lib()
data = datagen.KINDS[kind]((mib << 20) + 123, seed=9)
L.lrzgpu_profile_reset()
t0 = time.time()
s0, s1, st, crc, vr = B.hash_search(data, level=level)
dt = time.time() - t0
p = bench.Profile()
L.lrzgpu_profile_get(C.byref(p))
d = [int(v) for v in p.resolve_dbg]
print("%s %d MiB L%d: wall %.3f s, k_resolve %.1f ms in %d launches, lookups %d" % (kind, mib, level, dt, p.resolve_ms, p.resolve_launches, p.resolve_lookups), file=sys.stderr)
print("rounds %d committed %d exact steps %d stops: complex %d conflict %d" % (d[0], d[1], d[2], d[3], d[5]), file=sys.stderr)
lab = {8: "refill + publish", 9: "top-up + simulations", 4: "in-order pass", 10: "prefix counts + sweep", 14: "stops + filter writes", 15: "filter reads",
       7: "suspects published", 11: "suspects' exact test", 12: "table writes", 13: "bookkeeping + exact step + shift"}
tot = sum(d[k] for k in lab) or 1
for k in (8, 9, 4, 10, 14, 15, 7, 11, 12, 13):
    print("  %-34s %14d ticks %5.1f %% %9.1f per round" % (lab[k], d[k], 100.0 * d[k] / tot, d[k] / max(d[0], 1)), file=sys.stderr)
