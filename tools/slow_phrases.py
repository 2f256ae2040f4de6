#!/usr/bin/env python3
"""Where the time of the 'phrases' / 'few' test inputs goes (GPU box)."""
import os, sys, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import datagen
from conftest import load_bindings
import importlib.util
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py")); bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
B = load_bindings(); L = B.lib()
L.lrzgpu_profile_get.argtypes = [C.POINTER(bench.Profile)]
RAM = 80 * 100 * 1048576
for kind in ("phrases", "few", "text"):
    data = datagen.KINDS[kind](5 * 1048576 + 123, seed=9)
    for level in (7, 9):
        L.lrzgpu_profile_reset()
        t0 = time.time()
        img, _ = B.compress_buffer(data, level=level, threads=4, processors=8, ramsize=RAM, host_threads=8)
        dt = time.time() - t0
        p = bench.Profile(); L.lrzgpu_profile_get(C.byref(p))
        t0 = time.time(); B.decompress_buffer(img, host_threads=8); dd = time.time() - t0
        print("%s L%d: compress %.2f s (resolve %.0f ms, k_bt %.0f ms, finder %.0f ms, lz4 %.0f ms; enc busy %.1f s) decode %.2f s; image %d"
              % (kind, level, dt, p.resolve_ms, p.mf_bt_ms, p.mf_total_ms, p.lz4_ms, p.pipeline_s[0], dd, len(img)), flush=True)
        print("      resolver: lookups %d inserts %d; %s" % (p.resolve_lookups, p.resolve_inserts, dict(zip(
            ("batches", "committed", "serial_steps", "stop_complex", "stop_match", "stop_conflict", "stop_novictim", "stop_sweptrange"),
            [int(v) for v in p.resolve_dbg[:8]]))), flush=True)
    for wm in ("1000000000", "4096"):
        os.environ["LRZGPU_BT_MIN"] = wm
        L.lrzgpu_profile_reset()
        t0 = time.time()
        gc, gp = B.lzma_match_lists(data, dict_size=1 << 27, fb=64, cut=48, per_pos=110)
        p = bench.Profile(); L.lrzgpu_profile_get(C.byref(p))
        print("   lists alone wave_min %s: %.2f s, k_bt %.0f ms, wave dbg %s" % (wm, time.time() - t0, p.mf_bt_ms, list(p.mf_wave_dbg)), flush=True)
    os.environ.pop("LRZGPU_BT_MIN")
