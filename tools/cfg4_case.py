#!/usr/bin/env python3
"""BASELINE config 4 shape at 1.7 GiB (tar of 36 copies of a 48 MiB source tree, --zstd --zstd-level 15 -w 7): where the time goes."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import datagen, importlib.util
from conftest import load_bindings
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py")); bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
B = load_bindings(); L = B.lib(); L.lrzgpu_profile_get.argtypes = [C.POINTER(bench.Profile)]
data = datagen.source_tree_tar(36, 48 << 20, seed=7)
ram = os.sysconf("SC_PHYS_PAGES") * os.sysconf("SC_PAGE_SIZE")
for rep in range(2):
    L.lrzgpu_profile_reset()
    t0 = time.time()
    got, ctl = B.compress_buffer(data, level=7, threads=256, processors=256, ramsize=ram, window=7, zstd=True, zstd_level=15, host_threads=16, gpu_slots=8)
    dt = time.time() - t0
p = bench.Profile(); L.lrzgpu_profile_get(C.byref(p))
print("cfg4 %d MiB: %.2f s = %.1f MB/s, image %d" % (len(data) >> 20, dt, (len(data) >> 20) / dt, len(got)))
print("resolver: %.1f s summed in %d launches (union %.1f s); lookups %d inserts %d match bytes %d; %s" % (p.resolve_ms / 1e3, p.resolve_launches, p.union_ms[1] / 1e3, p.resolve_lookups, p.resolve_inserts, p.resolve_match_bytes, dict(zip(
    ("batches", "committed", "serial_steps", "stop_complex", "stop_match", "stop_conflict", "stop_novictim", "stop_sweptrange"), [int(v) for v in p.resolve_dbg[:8]]))))
print("pipeline: enc busy %.1f idle %.1f, last scan %.1f, last enc %.1f, wall %.1f" % (p.pipeline_s[0], p.pipeline_s[1], p.pipeline_s[4], p.pipeline_s[6], p.pipeline_s[7]))
