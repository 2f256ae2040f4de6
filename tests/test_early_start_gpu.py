"""Early start of blocks in the whole-file driver (the reference's back end starts on a buffer
while the scan is still running, src/stream.c:1836-1875, and its encoder consumes match blocks as the finder produces
them, src/lzma/C/LzFindMt.c:946-981, LzmaEnc.c:1079-1123): a block under construction goes to the finder as a growing
PREFIX (lrzgpu_lzma_match_lists_prefix's mode) and to an encoder that follows the lists stage by stage; the gate's
verdict is taken for granted and a refusal withdraws the block.  None of it may change a byte: every case here is the
whole .lrz image against the oracle's, with the early start FORCED on every block (LRZGPU_EARLY_START=2; by default it
only happens while encoders have nothing to do, i.e. in every other test's first moments too) and small steps, so that
a block sees many finder runs."""
import ctypes as C
import hashlib

import numpy as np
import pytest

import datagen
from test_compress_gpu import _both, RAM

pytestmark = pytest.mark.gpu


@pytest.fixture
def forced(monkeypatch):
    monkeypatch.setenv("LRZGPU_EARLY_START", "2")
    monkeypatch.setenv("LRZGPU_EARLY_STEP", str(1 << 20))
    monkeypatch.setenv("LRZGPU_SEG_BYTES", str(1 << 20))  # scan progress (and with it a new piece of the block) every MiB
    monkeypatch.setenv("LRZGPU_CENSUS", "0")  # (noise would skip the scan, and its blocks would be there whole at once)
    return monkeypatch


def _profile(B):
    from test_chunks_gpu import _bench, _profile as prof
    return prof(B, _bench())


@pytest.mark.parametrize("kind", ["text", "longrange", "random", "few"])
def test_every_block_started_early(B, O, forced, kind):
    """34 MiB in 10 MiB blocks (-p16; 24 MiB for the copy-heavy kind): three whole blocks started at their first MiB and followed through ~10 finder
    runs each, and a short last block whose early start is withdrawn when the chunk ends (its length, hence the
    encoder's view of it, was a guess).  'random': the gate refuses every literal block AFTER its encoder started."""
    # (a four-symbol alphabet makes the slowest blocks there are for resolver, finder and parser alike: one whole block + the
    #  short last one of those -- 50 repeated phrases, as slow, go through the same paths in tests/test_compress_gpu.py --;
    #  three of the others)
    n = ((34 << 20) + 77 if kind != "longrange" else (24 << 20) + 77) if kind not in ("phrases", "few") else (10 << 20) + (640 << 10) + 5
    data = datagen.KINDS[kind](n, seed=41)
    if kind in ("phrases", "few"):
        forced.setenv("LRZGPU_EARLY_STEP", str(3 << 20))  # (a finder run on such a prefix takes seconds: four of them, not ten)
    B.lib().lrzgpu_profile_reset()
    fs = _both(B, O, data, level=7, threads=16, processors=16)
    assert fs.stream_bufsize == 10 << 20
    p = _profile(B)
    if kind in ("text", "random"):  # (little is matched away: stream 1 really has those three blocks)
        assert p.early_s[2] >= 3 and p.early_s[3] >= 3 * 5, list(p.early_s)


def test_gate_refuses_after_the_encoder_started(B, O, forced, capfd):
    """The verdict of the lz4 gate is taken for granted when a block is started early, after a look at its first part:
    blocks of 64 KiB of text followed by random bytes pass that look (lz4 saves 20 KB on the first MiB) and are refused
    as a whole (20 KB saved, 40 KB of lz4's own overhead on 10 MiB of noise) -- with an encoder inside.  The block is
    stored, like the oracle's; the trace shows that the late refusal really happened."""
    forced.setenv("LRZGPU_TRACE", "2")
    parts = []
    for k in range(3):
        parts.append(datagen.text_like(64 << 10, seed=70 + k))
        parts.append(datagen.random_bytes((10 << 20) - (64 << 10), seed=80 + k))
    parts.append(datagen.random_bytes((2 << 20) + 11, seed=90))
    data = b"".join(parts)
    fs = _both(B, O, data, level=7, threads=16, processors=16)
    assert fs.stream_bufsize == 10 << 20
    err = capfd.readouterr().err
    assert " refused_late " in err, "no block was refused after its encoder had started"


def test_list_array_outgrown_while_an_encoder_reads_it(B, O, forced, capfd):
    """The host array of an early block's lists is sized from the density of its first finder run (x 1.5 + 4 M words).
    Blocks that begin with a MiB holding few pairs (units of eight random bytes written twice: enough for the gate's
    look at the first 256 KiB to say "compressible", nothing for rzip, 0.4 pairs per position) and go on as text
    (2.3 pairs per position: 21 M words for the 9 MiB, against an array of ~10 M) outgrow it a few runs later, while the
    encoder that was offered the block after the first run is reading the old array: the new one must be complete
    before the encoder is handed it (it used to be swapped in empty and filled afterwards)."""
    forced.setenv("LRZGPU_EARLY_STEP", str(1 << 19))
    forced.setenv("LRZGPU_TRACE", "2")
    parts = []
    for k in range(3):
        r = np.frombuffer(datagen.random_bytes(1 << 19, seed=300 + k), dtype=np.uint8).reshape(-1, 8)
        head = np.concatenate([r, r], axis=1).tobytes()
        parts.append(head)
        parts.append(datagen.text_like((10 << 20) - len(head), seed=320 + k))
    parts.append(datagen.text_like((3 << 20) + 17, seed=330))
    data = b"".join(parts)
    B.lib().lrzgpu_profile_reset()
    fs = _both(B, O, data, level=7, threads=16, processors=16)
    assert fs.stream_bufsize == 10 << 20
    p = _profile(B)
    assert p.early_s[2] >= 3 and p.early_s[3] >= 3 * 8, list(p.early_s)
    assert " lists_regrown " in capfd.readouterr().err, "no block outgrew its list array"


@pytest.mark.parametrize("level", [1, 4, 6, 9])
def test_levels_started_early(B, O, forced, level):
    """HC5 lists + the greedy parser (levels 1-4) and the other dictionaries / fast-byte settings behind the same path."""
    data = datagen.text_like((13 << 20) + 1234, seed=50 + level)  # (one whole 10 MiB block + a short one)
    _both(B, O, data, level=level, threads=16, processors=16)


def test_step_sizes(B, O, forced):
    data = datagen.long_range((26 << 20) + 9, seed=43, base_frac=0.2, mutate_every=70001)
    for step, seg in ((4096, 1 << 16), (300000, 1 << 18), (5 << 20, 1 << 20)):
        forced.setenv("LRZGPU_EARLY_STEP", str(step))
        forced.setenv("LRZGPU_SEG_BYTES", str(seg))
        _both(B, O, data[:(12 << 20) + 3] if step == 4096 else data, level=7, threads=16, processors=16)


def test_rollback_of_blocks_started_early(B, O, forced):
    """A later match reaching back over literal bytes already released voids the chunk's early blocks -- including the
    one an encoder is following (tests/test_compress_gpu.py test_early_release_rollback's second case, forced early)."""
    forced.setenv("LRZGPU_SPEC_MARGIN", "0")
    forced.setenv("LRZGPU_SEG_BYTES", str(4 << 20))
    base = datagen.random_bytes(4 << 20, seed=81)
    parts, pos = [base], len(base)
    for k in range(14, 18):
        start = k * (4 << 20) - (1 + k % 2)
        parts.append(datagen.text_like(start - pos, seed=100 + k))
        parts.append(base[:65536])
        pos = start + 65536
    parts.append(datagen.text_like(1 << 20, seed=99))
    data = b"".join(parts)
    B.lib().lrzgpu_profile_reset()
    _both(B, O, data, level=7, threads=16, processors=16)
    p = _profile(B)
    assert p.spec_rollbacks >= 1 and p.spec_cancelled_blocks >= 1


def test_multi_chunk_default_steps(B, O, monkeypatch):
    """Three 100 MiB chunks, 71.7 MB blocks, the default step (1/16 block), chunks scanned side by side: every first
    block is followed by an encoder from its first sixteenth on."""
    monkeypatch.setenv("LRZGPU_EARLY_START", "2")
    data = datagen.long_range(250 * 1048576 + 4097, seed=14, base_frac=0.08, mutate_every=300007)
    B.lib().lrzgpu_profile_reset()
    fs = _both(B, O, data, level=7, threads=4, processors=8, window=1)
    assert fs.n_chunks == 3
    p = _profile(B)
    assert p.early_s[2] >= 3


def test_victim_round_rescan_with_early_blocks(B, O, forced):
    """A chunk scanned from the wrong victim_round is voided and scanned again -- with encoders inside its blocks."""
    from test_chunks_gpu import _moving_victim_data
    data = _moving_victim_data(O)
    want, fs = O.compress_buffer(data, compression_level=7, threads=8, processors=16, ramsize=RAM, window=1, workers=16)
    B.lib().lrzgpu_profile_reset()
    got, ctl = B.compress_buffer(data, level=7, threads=8, processors=16, ramsize=RAM, window=1, host_threads=16)
    assert got == want
    assert _profile(B).victim_rescans >= 1


def test_stdin_mode_and_device_input_started_early(B, O, forced):
    import torch
    ram = 60 << 20
    chunk = (ram // 3) // 4096 * 4096
    data = (datagen.text_like(chunk, seed=5) * 3)[:2 * chunk + 12345]
    want, _ = O.compress_buffer(data, compression_level=1, threads=2, processors=2, ramsize=ram, workers=4, stdin_mode=1, stdout_mode=1)
    got, _ = B.compress_buffer(data, level=1, threads=2, processors=2, ramsize=ram, host_threads=4, stdin_mode=1, stdout_mode=1)
    assert got == want
    data = datagen.long_range((31 << 20) + 5, seed=15, base_frac=0.3)
    want, _ = O.compress_buffer(data, compression_level=7, threads=16, processors=16, ramsize=RAM, workers=8)
    t = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
    got, ctl = B.compress_device(t.data_ptr(), t.numel(), level=7, threads=16, processors=16, ramsize=RAM, host_threads=8)
    assert got == want and bytes(ctl.hash_resblock) == hashlib.md5(data).digest()


def test_early_start_off_is_the_same_image(B, O, monkeypatch):
    data = datagen.text_like((21 << 20) + 3, seed=77)
    monkeypatch.setenv("LRZGPU_EARLY_START", "0")
    B.lib().lrzgpu_profile_reset()
    _both(B, O, data, level=7, threads=16, processors=16)
    assert _profile(B).early_s[2] == 0


# ---- the pieces underneath: the finder on a prefix of a block, the parser on lists that arrive in stages ----------
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
        for step in (0, 100000):  # the rest in one piece / in ten
            rc, got = B.lzma_encode_with_lists_staged(data, counts, lists, P - fb - 4, level=level, dict_size=dict_size, fb=fb,
                                                      list_format=fmt, early_counts=pcounts, early_pairs=plists, stage_step=step)
            assert rc == 0 and got == want, (fmt, step)


