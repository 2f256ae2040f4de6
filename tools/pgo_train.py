#!/usr/bin/env python3
"""Training workload for the profile-guided build of the host parser (run with an instrumented liblrzgpu.so, see
tools/pgo_record.sh): the bench text through the optimal parser in the pipeline's list format (most weight; part of it
with the lists arriving in stages, as an early-started block sees them), the other list formats, the other test kinds,
and the greedy parser of levels 1-4.  Match lists come from the GPU finder when a GPU is visible, else from the oracle's
restated reference finder (same lists; a profile is branch counts, it does not depend on the CPU it was recorded on)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import datagen
from conftest import load_bindings
B = load_bindings()
GPU = B.lib().lrzgpu_device_count() > 0
if not GPU:
    import numpy as np
    import oracle_lib as O

def run(data, level, fmts, dict_size, fb):
    if not GPU:
        offs, pairs = (O.mf_bt4(data, dict_size=dict_size, fb=fb, cut=16 + fb // 2) if level >= 5 else
                       O.mf_hc5(data, dict_size=dict_size, fb=fb, cut=(16 + fb // 2) // 2))
        counts = np.diff(offs).astype(np.uint8)
    elif level >= 5:
        counts, pairs = B.lzma_match_lists(data, dict_size=dict_size, fb=fb, cut=16 + fb // 2, per_pos=24)
    else:
        counts, pairs = B.lzma_match_lists_hc5(data, dict_size=dict_size, fb=fb, cut=(16 + fb // 2) // 2)
    for fmt in fmts:
        lists = B.format_lists(data, counts, pairs, fmt) if fmt else pairs
        rc, out = B.lzma_encode_with_lists(data, counts, lists, level=level, dict_size=dict_size, fb=fb, list_format=fmt)
        assert rc == 0, (level, fmt, rc)
    if len(data) >= (8 << 20):  # the staged entry (early start): lists handed over an eighth of the block at a time
        fmt = fmts[0]
        lists = B.format_lists(data, counts, pairs, fmt) if fmt else pairs
        rc, out = B.lzma_encode_with_lists_staged(data, counts, lists, len(data) // 8, level=level, dict_size=dict_size, fb=fb,
                                                  list_format=fmt, stage_step=len(data) // 8)
        assert rc == 0, (level, fmt, rc, "staged")

text = datagen.text_alnum(16 << 20, seed=1)
run(text, 7, (2, 2, 2, 1, 0), 1 << 25, 64)
run(text[:6 << 20], 5, (2,), 1 << 24, 32)
run(text[:6 << 20], 9, (0,), 1 << 27, 64)
for kind in ("random", "phrases", "sparse", "zeros", "longrange"):
    d = datagen.KINDS[kind](3 << 20, seed=3)
    run(d, 7, (2, 0), 1 << 25, 64)
    run(d, 3, (2, 0), 1 << 20, 32)
run(text[:8 << 20], 3, (2, 2, 0), 1 << 22, 32)
run(text[:4 << 20], 1, (2,), 1 << 18, 32)
print("training workload done")
