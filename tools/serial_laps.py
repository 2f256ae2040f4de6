#!/usr/bin/env python3
"""Sections of the exact steps of a serial episode: needs tools/experiments/resolver_serial_steps_as_functions.patch applied and a build
with -DLRZGPU_SERIAL_LAPS (make HIPFLAGS_rzip_scan="... -DLRZGPU_SERIAL_LAPS"); ticks of s_memtime.
usage: python tools/serial_laps.py [few|phrases] [level]"""
import os, sys, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import datagen
from conftest import load_bindings
import importlib.util
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py")); bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
B = load_bindings(); L = B.lib()
kind = sys.argv[1] if len(sys.argv) > 1 else "few"
level = int(sys.argv[2]) if len(sys.argv) > 2 else 7
data = datagen.KINDS[kind](5 * 1048576 + 123, seed=9)
B.hash_search(data[:1 << 20], level=level)
L.lrzgpu_profile_reset()
t0 = time.time()
B.hash_search(data, level=level)
dt = time.time() - t0
p = bench.Profile(); L.lrzgpu_profile_get(C.byref(p))
d = [int(v) for v in p.resolve_dbg]
print("%s L%d: scan %.2f s, k_resolve %.0f ms, lookups %d" % (kind, level, dt, p.resolve_ms, p.resolve_lookups))
print("  rounds %d committed %d exact steps %d in %d episodes" % (d[0], d[1], d[2], d[13]))
ms = p.resolve_ms / max(d[15], 1)  # ms per tick, from the kernel's own duration
print("  kernel %d ticks (%.2f GHz)" % (d[15], d[15] / p.resolve_ms / 1e6))
for k, n in ((8, "light step: the probe window"), (9, "light step: keep or emit"), (12, "episodes, whole (call included)")):
    print("  %-32s %8.1f ms  %6.0f ticks per exact step" % (n, d[k] * ms, d[k] / max(d[2], 1)))
print("  %-32s %8.1f ms  %6.0f ticks per round" % ("everything else (rounds)", (d[15] - d[12]) * ms, (d[15] - d[12]) / max(d[0], 1)))
