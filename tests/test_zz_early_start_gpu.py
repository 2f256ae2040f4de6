"""Tests written when no GPU time was left in their round (named to run last, so that a surprise costs only themselves):
the GPU half of the early start of a block as far as it exists -- the BT4 finder on a
prefix of a block with the block's hash mask, and its lists driving the two-stage parser -- and the finder's bucket
kernels on two streams."""
import numpy as np
import pytest

import datagen

pytestmark = pytest.mark.gpu


def _lists_from_oracle(O, data, dict_size, fb, cut):
    offs, pairs = O.mf_bt4(data, dict_size=dict_size, fb=fb, cut=cut)
    return np.diff(offs).astype(np.uint8), pairs


@pytest.mark.parametrize("kind", ["text", "few"])
def test_prefix_run_of_the_finder_gives_the_blocks_lists(B, O, kind):
    """A prefix shorter than the dictionary under a block longer than it: the hash mask derived from the prefix's own
    length would be a different one (2^18 - 1 against 2^19 - 1), so the buckets -- and, wherever the cut-off counts
    visits, the lists -- are the block's only because the finder is told the block's size.  Oracle = the restated
    reference finder on the WHOLE block."""
    n, P, fb, dict_size = 1200000, 400000, 64, 1 << 20
    data = datagen.KINDS[kind](n, seed=41)
    cut = 16 + fb // 2
    offs, pairs = O.mf_bt4(data, dict_size=dict_size, fb=fb, cut=cut)
    gc, gp = B.lzma_match_lists_prefix(data[:P], n, dict_size=dict_size, fb=fb, cut=cut, per_pos=110)
    safe = P - fb - 4
    want_counts = np.diff(offs).astype(np.uint8)
    assert np.array_equal(gc[:safe], want_counts[:safe])
    assert np.array_equal(gp[:int(offs[safe])], pairs[:int(offs[safe])])


def test_early_start_with_lists_from_the_gpu(B, O):
    """Prefix lists and whole-block lists both from the GPU finder, the two-stage parser on top: reference bytes."""
    n, P, fb, dict_size, level = 1500000, 500000, 64, 1 << 20, 7
    data = datagen.text_like(n, seed=43)
    cut = 16 + fb // 2
    rc, want, _ = O.lzma_compress_ref(data, level=level, dict_size=dict_size)
    assert rc == 0
    counts, pairs = B.lzma_match_lists(data, dict_size=dict_size, fb=fb, cut=cut, per_pos=110)
    pcounts, ppairs = B.lzma_match_lists_prefix(data[:P], n, dict_size=dict_size, fb=fb, cut=cut, per_pos=110)
    for fmt in (0, 2):
        lists = B.format_lists(data, counts, pairs, fmt)
        plists = B.format_lists(data[:P], pcounts, ppairs, fmt)
        rc, got = B.lzma_encode_with_lists_staged(data, counts, lists, P - fb - 4, level=level, dict_size=dict_size, fb=fb,
                                                  list_format=fmt, early_counts=pcounts, early_pairs=plists)
        assert rc == 0 and got == want, fmt


@pytest.mark.parametrize("wave_min,lds_min", [("64", "0"), ("2048", "0"), ("4096", "300")])
def test_match_lists_with_the_bucket_kernels_side_by_side(B, O, wave_min, lds_min):
    """LRZGPU_BT_OVERLAP=1: the wave-per-bucket launches (tree in memory, and in LDS with lds_min) run on a second
    stream beside the lane-per-bucket kernel; they share the pool cursor only.  Same lists."""
    import os
    keys = ("LRZGPU_BT_OVERLAP", "LRZGPU_BT_WAVE_MIN", "LRZGPU_BT_LDS_MIN")
    old = {k: os.environ.get(k) for k in keys}
    os.environ.update({"LRZGPU_BT_OVERLAP": "1", "LRZGPU_BT_WAVE_MIN": wave_min, "LRZGPU_BT_LDS_MIN": lds_min})
    try:
        cases = [
            (datagen.text_alnum(1200000, seed=6, nwords=400) + datagen.text_alnum(800000, seed=8, nwords=60), 1 << 25, 64),
            (datagen.KINDS["few"](400000, seed=32), 1 << 25, 64),
            (datagen.phrase_mix(500000, seed=33), 1 << 16, 32),
            (datagen.text_like(3000, seed=34) + bytes(100000) + b"\x07" * 2500 + b"ab" * 1800, 1 << 25, 64),
        ]
        for data, dict_size, fb in cases:
            cut = 16 + fb // 2
            oc, op = _lists_from_oracle(O, data, dict_size, fb, cut)
            gc, gp = B.lzma_match_lists(data, dict_size=dict_size, fb=fb, cut=cut, per_pos=110)
            assert np.array_equal(gc, oc), (wave_min, lds_min, len(data), dict_size)
            assert np.array_equal(gp, op), (wave_min, lds_min, len(data), dict_size)
    finally:
        for k in keys:
            if old[k] is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = old[k]
