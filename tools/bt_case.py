#!/usr/bin/env python3
"""Match finder alone on one block of the bench text (GPU): k_bt / k_bt_group timing per cut-over, pipeline statistics.
Arguments: MiB, then settings "WAVE_MIN" or "WAVE_MIN,GROUP_MIN" (LRZGPU_BT_MIN=wave,group: buckets from this length on get a
wavefront; from this length on eight lanes; below: one lane); the lists of every setting are
compared with the first one's."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import importlib.util
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py")); bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
B = bench.load_bindings(); L = B.lib()
L.lrzgpu_profile_get.argtypes = [C.POINTER(bench.Profile)]
mib = int(sys.argv[1]) if len(sys.argv) > 1 else 64
data = bench.text_like_torch(mib << 20, 1, torch.device("cuda:0")).cpu().numpy().tobytes()
import numpy as np
first = None
for setting in sys.argv[2:] or ["1000000000", "1024"]:
    f = setting.split(",")
    wm, lm = f[0], (f[1] if len(f) > 1 else f[0])
    os.environ["LRZGPU_BT_MIN"] = "%s,%s" % (wm, lm)
    for rep in range(int(os.environ.get("BT_CASE_REPS", "2"))):
        L.lrzgpu_profile_reset()
        t0 = time.time()
        gc, gp = B.lzma_match_lists(data, dict_size=1 << 25, fb=64, cut=48, per_pos=16)
        dt = time.time() - t0
        p = bench.Profile(); L.lrzgpu_profile_get(C.byref(p))
    d = list(p.mf_wave_dbg)
    if first is None:
        first = (gc, gp)
    else:
        print("  lists equal to the first setting's:", bool(np.array_equal(gc, first[0]) and np.array_equal(gp, first[1])))
    print("wave_min,group_min %s: k_bt %.1f ms, finder %.1f ms, wall %.2f s; pipelined kernels: %d positions, rounds %d, visits %d (%.1f/pos), waits %d, visits/round %.2f"
          % (setting, p.mf_bt_ms, p.mf_total_ms, dt, d[3], d[0], d[1], d[1] / max(d[3], 1), d[2], d[1] / max(d[0], 1)), flush=True)
