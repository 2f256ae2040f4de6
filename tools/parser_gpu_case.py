#!/usr/bin/env python3
"""Host parser speed on the GPU box's CPU: lists from the GPU finder, then T concurrent parser threads
(one block each, like the pipeline's encoders) for each list format.  Shows what the encoders can do per
thread and how they scale against the cgroup CPU quota / shared caches."""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import datagen
from conftest import load_bindings

keep = []

def main():
    B = load_bindings()
    sizes = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "16,64").split(",")]
    threads = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "1,8,16").split(",")]
    fmts = [int(x) for x in (sys.argv[3] if len(sys.argv) > 3 else "0,2").split(",")]
    for mib in sizes:
        n = mib << 20
        data = datagen.text_alnum(n, seed=1)
        if os.environ.get("HUGE"):
            import mmap, ctypes
            mm = mmap.mmap(-1, (n + (4 << 20)) & ~((2 << 20) - 1))
            mm.madvise(mmap.MADV_HUGEPAGE)
            mm[:n] = data
            keep.append(mm)
            data = (ctypes.c_char * n).from_buffer(mm)
        t0 = time.time()
        counts, pairs = B.lzma_match_lists(data, dict_size=1 << 25, fb=64, cut=48, per_pos=12)
        print("%d MiB: GPU finder (with H2D/D2H, fresh workspace) %.2f s, %.2f pairs/pos" % (mib, time.time() - t0, len(pairs) / 2 / n), flush=True)
        for fmt in fmts:
            lists = B.format_lists(data, counts, pairs, fmt)
            for T in threads:
                res = [None] * T
                def work(i):
                    t = time.time()
                    rc, out = B.lzma_encode_with_lists(data, counts, lists, level=7, dict_size=1 << 25, fb=64, list_format=fmt)
                    res[i] = (time.time() - t, rc, len(out))
                th = [threading.Thread(target=work, args=(i,)) for i in range(T)]
                t0 = time.time()
                for t in th: t.start()
                for t in th: t.join()
                wall = time.time() - t0
                per = [r[0] for r in res]
                print("  fmt %d  T=%2d: wall %.2f s  per-thread %.2f..%.2f s  = %.2f MiB/s per thread, %.1f MiB/s total (rc %d, %d B)"
                      % (fmt, T, wall, min(per), max(per), mib / (sum(per) / T), T * mib / wall, res[0][1], res[0][2]), flush=True)

main()
