#!/usr/bin/env python3
"""lz4 gate kernel alone on one 64 MiB block of the bench text: full size vs the early verdict (GPU box)."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, importlib.util
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py")); bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
B = bench.load_bindings(); L = B.lib()
n = 64 << 20
data = bench.text_like_torch(n, 1, torch.device("cuda:0")).cpu().numpy().tobytes()
L.lrzgpu_lz4_size_stop_below.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int]
for stop in (0, n, n // 2):
    for rep in range(2):
        t0 = time.time()
        r = L.lrzgpu_lz4_size_stop_below(data, n, n + 1, stop, 0)
        dt = time.time() - t0
    print("stop_below %d: result %d (%.1f %% of the block) in %.1f ms including the upload" % (stop, r, 100.0 * r / n, dt * 1e3), flush=True)
