"""The reference's stream.h surface (prepare_streamout_threads / open_stream_out / write_stream / flush_buffer /
close_stream_out / close_streamout_threads) and the per-block dispatch seam (lzma_compress_buf contract),
driven the way rzip.c drives them: the chunks + blocks they write must be the bytes of the whole-file path."""
import ctypes as C

import pytest

import datagen

pytestmark = pytest.mark.gpu

RAM = 80 * 100 << 20


def _chunks(B, O, data, window, level=7):
    """rzip streams per chunk (from the GPU scan, already proven equal to the oracle's elsewhere)."""
    plan = B.make_control(level=level, threads=4, processors=8, ramsize=RAM, window=window)
    cs = C.c_int64()
    assert B.lib().lrzgpu_plan(C.byref(plan), len(data), C.byref(cs)) == 0
    out, vr, off = [], 0, 0
    while True:
        n = min(cs.value, len(data) - off)
        s0, s1, st, crc, vr = B.hash_search(data[off:off + n], level=level, victim_round=vr)
        out.append((n, B.chunk_bytes_for(n), s0, s1))
        off += n
        if off >= len(data):
            return out


@pytest.mark.parametrize("kind,n,window", [("longrange", 6 * 1048576 + 77, 0), ("text", 3 * 1048576, 0), ("random", 2 * 1048576, 0),
                                           ("cfg3", 104857600 + 40 * 1048576 + 4321, 1), ("tiny", 100, 0), ("empty", 0, 0)])
def test_stream_api_writes_the_same_file(B, O, tmp_path, kind, n, window):
    if kind == "cfg3":
        data = datagen.cfg3(n, 15 * 1048576, seed=4)
    elif kind in ("tiny", "empty"):
        data = datagen.text_like(n, seed=2)
    else:
        data = datagen.KINDS[kind](n, seed=21)
    kw = dict(level=7, threads=4, processors=8, ramsize=RAM, window=window)
    want, _ = O.compress_buffer(data, compression_level=7, threads=4, processors=8, ramsize=RAM, window=window, workers=16)
    ctl, end = B.stream_out_file(str(tmp_path / "s.lrz"), _chunks(B, O, data, window), len(data), host_threads=8, **kw)
    got = (tmp_path / "s.lrz").read_bytes()
    assert end == len(got) == len(want) - 16          # everything but the MD5 trailer
    assert got[21:] == want[21:-16]                    # chunk headers, stream headers, chained blocks


def test_stream_api_no_compress_and_zstd(B, O, tmp_path):
    data = datagen.long_range(5 * 1048576 + 9, seed=31, base_frac=0.4)
    for extra, okw in ((dict(no_compress=True), dict(no_compress=1)), (dict(zstd=True, zstd_level=7), dict(zstd=1, zstd_level=7))):
        want, _ = O.compress_buffer(data, compression_level=6, threads=3, processors=8, ramsize=RAM, workers=8, **okw)
        rl = 0
        if "zstd" in extra:  # --zstd-level drives the rzip level (src/main.c:692-711): strategy 4 for level 7
            rl = 4
        chunks = _chunks(B, O, data, 0, level=rl or 6)
        ctl, end = B.stream_out_file(str(tmp_path / "z.lrz"), chunks, len(data), host_threads=8, level=6, threads=3, processors=8,
                                     ramsize=RAM, **extra)
        assert (tmp_path / "z.lrz").read_bytes()[21:] == want[21:-16]


def test_lzma_compress_buf_contract(B, O):
    """compressible -> s_buf replaced, c_type 6, bytes of the reference LzmaCompress; lz4 gate says no ->
    untouched, return 0; too short (< 64 is the caller's test) still passes through the gate."""
    text = datagen.text_like(2 * 1048576 + 13, seed=41)
    rc, c_type, out = B.lzma_compress_buf(text, level=7, threads=4, processors=8, ramsize=RAM)
    r2, want, _ = O.lzma_compress_ref(text, level=7, dict_size=1 << 25)
    assert (rc, c_type) == (0, 6) and r2 == 0 and out == want
    rnd = datagen.random_bytes(1 << 20, seed=42)
    rc, c_type, out = B.lzma_compress_buf(rnd, level=7, threads=4, processors=8, ramsize=RAM)
    assert (rc, c_type) == (0, 3) and out == rnd
    rc, c_type, out = B.lzma_compress_buf(text[:300000], level=3, threads=4, processors=8, ramsize=RAM)
    r2, want, _ = O.lzma_compress_ref(text[:300000], level=3, dict_size=1 << 22)
    assert (rc, c_type) == (0, 6) and out == want
