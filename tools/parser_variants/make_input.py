#!/usr/bin/env python3
"""Writes the harness's input: a block of the bench text, list counts and packed lists (GPU finder when a GPU is
visible, else the oracle's restated finder -- same lists)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import datagen
from conftest import load_bindings
B = load_bindings()
mib = int(sys.argv[1]) if len(sys.argv) > 1 else 64
out = sys.argv[2] if len(sys.argv) > 2 else "/tmp/pv"
os.makedirs(out, exist_ok=True)
n = mib << 20
data = datagen.text_alnum(n, seed=1)
if B.lib().lrzgpu_device_count() > 0:
    counts, pairs = B.lzma_match_lists(data, dict_size=1 << 25, fb=64, cut=48, per_pos=12)
else:
    import oracle_lib as O
    offs, pairs = O.mf_bt4(data, dict_size=1 << 25, fb=64, cut=48)
    counts = np.diff(offs).astype(np.uint8)
open(os.path.join(out, "d.bin"), "wb").write(data)
counts.tofile(os.path.join(out, "c.bin"))
B.format_lists(data, counts, pairs, 2).tofile(os.path.join(out, "p2.bin"))
print("input written: %d MiB, %.2f pairs per position" % (mib, len(pairs) / 2 / n))
