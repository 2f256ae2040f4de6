"""The GPU half of the early start of a block as far as it exists: the BT4 finder on a
prefix of a block with the block's hash mask, and its lists driving the two-stage parser.  (Named to run last: it was
written when no GPU time was left in its round.)"""
import numpy as np
import pytest

import datagen

pytestmark = pytest.mark.gpu


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
