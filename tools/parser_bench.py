#!/usr/bin/env python3
"""Host LZMA parser micro-benchmark (no GPU needed): match lists from the oracle finder, then the product's
parser + range coder alone (lrzgpu_lzma_encode_with_lists), checked against oracle/_ref LzmaCompress."""
import os, sys, time, pickle
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import datagen, oracle_lib as O
from conftest import load_bindings

def main():
    mib = float(sys.argv[1]) if len(sys.argv) > 1 else 8
    level = int(sys.argv[2]) if len(sys.argv) > 2 else 7
    kind = sys.argv[3] if len(sys.argv) > 3 else "alnum"
    reps = int(sys.argv[4]) if len(sys.argv) > 4 else 1
    n = int(mib * 1048576)
    B = load_bindings()
    cache = "/tmp/parser_bench_%s_%d_%d.pkl" % (kind, n, level)
    fb = 32 if level < 7 else 64
    dict_size = {5: 1 << 24, 6: 1 << 25, 7: 1 << 25, 8: 1 << 26, 9: 1 << 27}[level]
    if os.path.exists(cache):
        data, counts, pairs, want = pickle.load(open(cache, "rb"))
    else:
        data = datagen.text_alnum(n, seed=1) if kind == "alnum" else datagen.KINDS[kind](n, seed=1)
        offs, pairs = O.mf_bt4(data, dict_size=dict_size, fb=fb, cut=16 + fb // 2)
        counts = np.diff(offs).astype(np.uint8)
        rc, want, _ = O.lzma_compress_ref(data, level=level, dict_size=dict_size)
        assert rc == 0
        pickle.dump((data, counts, pairs, want), open(cache, "wb"))
    fmt = int(os.environ.get("LIST_FORMAT", "0"))
    if fmt:
        pairs = B.format_lists(data, counts, pairs, fmt)
    best = 1e9
    for _ in range(reps):
        t0 = time.time()
        rc, got = B.lzma_encode_with_lists(data, counts, pairs, level=level, dict_size=dict_size, fb=fb, list_format=fmt)
        best = min(best, time.time() - t0)
    print("fmt %d " % fmt, end="")
    print("%s %.1f MiB L%d: %.3f s  %.2f MiB/s  %s (%d -> %d bytes, %.2f pairs/pos)" % (
        kind, mib, level, best, mib / best, "BIT-EXACT" if (rc == 0 and got == want) else "MISMATCH rc=%d" % rc, n, len(got), len(pairs) / 2 / n))

main()
