"""GPU parity of the rzip scan (tags, hash-table automaton, match extension, literal gather, CRC)
against the oracle restatement of src/rzip.c hash_search, through the C ABI."""
import zlib

import pytest

import datagen

pytestmark = pytest.mark.gpu


def _check(B, O, data, level, victim_round=0):
    o0, o1, ost, ocrc, ovr = O.rzip_chunk(data, level=level, victim_round=victim_round)
    g0, g1, gst, gcrc, gvr = B.hash_search(data, level=level, victim_round=victim_round)
    assert gcrc == ocrc == (zlib.crc32(data) & 0xFFFFFFFF)
    assert g0 == o0, ("stream0", len(g0), len(o0))
    assert g1 == o1, ("stream1", len(g1), len(o1))
    assert gvr == ovr
    for f in ("matches", "match_bytes", "literals", "literal_bytes", "inserts", "lookups", "tag_hits", "tag_misses",
              "hash_count", "tag_clean_ptr", "minimum_tag_mask", "tag_mask"):
        assert getattr(gst, f) == getattr(ost, f), f
    return gst


@pytest.mark.parametrize("n", [0, 1, 30, 31, 32, 61, 62, 63, 100, 4095, 4096, 4097, 70000])
@pytest.mark.parametrize("kind", ["text", "zeros", "few", "longrange"])
def test_tiny_chunks(B, O, kind, n):
    _check(B, O, datagen.KINDS[kind](n, seed=3), level=7)


@pytest.mark.parametrize("level", [1, 4, 6, 7, 9])
@pytest.mark.parametrize("kind", ["text", "random", "few", "phrases", "sparse", "zeros", "longrange"])
def test_scan_equals_oracle(B, O, kind, level):
    n = (1 if kind in ("phrases", "few") else 3) * 1048576 + 777  # (collapsed tag spaces: serial resolver steps, kept small)
    _check(B, O, datagen.KINDS[kind](n, seed=level + 11), level=level)


@pytest.mark.parametrize("dense", ["0", "1"])
@pytest.mark.parametrize("verdict", ["never", "always"])
@pytest.mark.parametrize("kind", ["few", "phrases", "text", "longrange", "random"])
def test_rounds_and_exact_stretches_are_one_automaton(B, O, kind, verdict, dense, monkeypatch):
    """The resolver replays hash_search (src/rzip.c:304-353, 495-534, 586-762) two ways: speculative rounds over a window of
    candidates, and stretches of exact steps when rounds stop paying (rzip_resolve_mw.h: a round is poor when it stopped
    early with fewer than eight candidates committed -- a count, not a clock).  LRZGPU_RESOLVE_POOR forces the verdict, so
    every data kind goes through BOTH paths on every box: 'never' = rounds only (the degenerate kinds commit two or three
    candidates a round: kept small), 'always' = a stretch after every eight rounds, doubling to 16 384 steps.
    LRZGPU_RESOLVE_DENSE=0 keeps the stretches inside the four-wavefront kernel (rounds 1 to 5); with 1 (the default) that
    kernel hands over to the dense variant instead, which takes its own stretches under 'always'."""
    monkeypatch.setenv("LRZGPU_RESOLVE_POOR", verdict)
    monkeypatch.setenv("LRZGPU_RESOLVE_DENSE", dense)
    if dense == "1" and (verdict == "never" or kind in ("longrange", "random")):
        pytest.skip("rounds only: the hand-over is never asked for, same run as with 0 (and three kinds are enough for it)")
    degenerate = kind in ("few", "phrases")
    n = ((96 << 10) if verdict == "never" else (1 << 20)) + 99 if degenerate else 1048576 + 777
    for level in ((7, 9) if degenerate and verdict == "always" else (7,)):
        _check(B, O, datagen.KINDS[kind](n, seed=9 + level), level=level)


@pytest.mark.parametrize("level", [1, 4, 6, 7, 9])
@pytest.mark.parametrize("kind", ["text", "random", "few", "phrases", "sparse", "zeros", "longrange"])
def test_dense_resolver_equals_oracle(B, O, kind, level, monkeypatch):
    """Every data kind through the DENSE variant of the resolver on every launch (LRZGPU_RESOLVE_DENSE=always; by default
    it only takes over where the four-wavefront rounds stop paying): one wavefront, real matches carried through the
    round by the in-order lazy-match pass (src/rzip.c:697-731), round-robin evictions of one tag passing each other
    (src/rzip.c:304-353).  Same streams, same statistics (tag hits and misses included) as the oracle."""
    monkeypatch.setenv("LRZGPU_RESOLVE_DENSE", "always")
    n = (2 if kind in ("phrases", "few") else 3) * 1048576 + 777
    if level == 9 and kind in ("phrases", "few"):
        n = 393216 + 777  # (chains of 128 equal tags are beyond a lane of the variant: exact steps, kept small)
    _check(B, O, datagen.KINDS[kind](n, seed=level + 11), level=level)


@pytest.mark.parametrize("dense", ["0", "always"])
def test_dense_resolver_tiny_and_ragged(B, O, dense, monkeypatch):
    monkeypatch.setenv("LRZGPU_RESOLVE_DENSE", dense)
    for kind in ("phrases", "few", "longrange", "zeros"):
        for n in (0, 31, 63, 100, 4097, 70000, 300001):
            _check(B, O, datagen.KINDS[kind](n, seed=5), level=7)


@pytest.mark.parametrize("kind", ["few", "phrases", "dna"])
def test_degenerate_inputs_hand_over_to_the_dense_resolver(B, O, kind):
    """Default switches: the four-wavefront resolver asks for the dense one when its rounds commit fewer than eight
    candidates (a four-letter alphabet: every candidate's insert lands in its neighbours' probe run; phrases: a match
    every few candidates), the dense one gives back when seven rounds in eight had no use for it.  4 MiB each (the
    reference's one core needs about 2 s for these; rounds 1 to 5 of this library needed 8 to 13 s for FIVE MiB)."""
    n = ((16 << 20) if kind == "few" else (4 << 20)) + 123  # (16 MiB of four letters: about 4 s here, 3.5 s for the oracle's one core)
    if kind == "dna":
        # four letters with repeats: a genome-like input (every 40 KB a copy of an earlier 1-3 KB stretch, lightly mutated)
        import numpy as np
        rng = np.random.default_rng(77)
        a = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=n)].copy()
        for at in range(50000, n - 4000, 40000):
            ln = int(rng.integers(1000, 3000))
            src = int(rng.integers(0, at - ln))
            a[at:at + ln] = a[src:src + ln]
            a[at + ln // 2] = 65 if a[at + ln // 2] != 65 else 67
        data = a.tobytes()
    else:
        data = datagen.KINDS[kind](n, seed=9)
    _check(B, O, data, level=7)


@pytest.mark.parametrize("kind,n", [("text", 5 * 1048576 + 12345), ("random", 3 * 1048576 + 31), ("few", 1048576 + 7), ("zeros", 70000),
                                    ("text", 4096 + 30), ("text", 4096 + 31), ("text", 31), ("text", 30), ("text", 0)])
def test_tag_candidates_equal_the_access_hooks(B, kind, n):
    """k_tag_scan on its own (lrzgpu_tag_candidates_dev): the positions whose 31-byte XOR tag (src/rzip.c:385-416) passes
    the mask, counted and checksummed on the device, against the tags recomputed here from hash_index[] -- a prefix XOR
    in numpy, tied to lrzgpu_full_tag / lrzgpu_next_tag (the scan access hooks) at sampled positions.
    Several masks, ragged starts (the kernel reads aligned 16-byte words; a segment may begin anywhere), tile edges."""
    import ctypes as C
    import numpy as np
    import torch
    data = datagen.KINDS[kind](n, seed=23)
    L = B.lib()
    hx = (C.c_uint64 * 256)()
    L.lrzgpu_hash_index(hx)
    hx = np.frombuffer(hx, dtype=np.uint64).copy()
    a = np.frombuffer(data, dtype=np.uint8)
    buf = torch.zeros(n + 256, dtype=torch.uint8, device="cuda")
    buf[:n] = torch.from_numpy(a.copy()).cuda()
    npos = max(0, n - 30)
    if npos:
        X = np.concatenate([[np.uint64(0)], np.bitwise_xor.accumulate(hx[a])])
        tags = X[31:31 + npos] ^ X[:npos]
        L.lrzgpu_full_tag.restype = C.c_uint64
        L.lrzgpu_full_tag.argtypes = [C.c_char_p, C.c_int64]
        L.lrzgpu_next_tag.restype = C.c_uint64
        L.lrzgpu_next_tag.argtypes = [C.c_char_p, C.c_int64, C.c_uint64]
        for p in sorted({0, min(1, npos - 1), npos // 3, npos - 1}):
            assert int(tags[p]) == L.lrzgpu_full_tag(data, p)
            if p:
                assert int(tags[p]) == L.lrzgpu_next_tag(data, p, int(tags[p - 1]))
    else:
        tags = np.zeros(0, dtype=np.uint64)
    pos = np.arange(npos, dtype=np.uint64)
    for mask in (1, 3, 0xF, 0x1FF):
        for first in (0, 1, 15, 16, 4097):
            sel = ((tags & np.uint64(mask)) == np.uint64(mask)) & (pos >= np.uint64(first))
            want_n = int(sel.sum())
            with np.errstate(over="ignore"):
                want = int((((pos[sel] + np.uint64(1)) * np.uint64(0x9E3779B97F4A7C15)) ^ (tags[sel] * np.uint64(0xC2B2AE3D27D4EB4F))).sum(dtype=np.uint64)) if want_n else 0
            got_n, got, _ = B.tag_candidates_dev(buf.data_ptr(), n, first=first, min_mask=mask)
            assert (got_n, got) == (want_n, want), (mask, first)


def test_table_fill_and_clean_sweeps(B, O):
    """Large enough that the L7 table (2.8M entries) fills and the clean sweep wraps several times."""
    n = 48 * 1048576 + 5
    st = _check(B, O, datagen.text_like(n, seed=21), level=7)
    assert st.minimum_tag_mask > 1
    # small table (level 1): many mask generations on a few MiB
    st = _check(B, O, datagen.random_bytes(6 * 1048576, seed=22), level=1)
    assert st.minimum_tag_mask > 15


def test_dense_resolver_through_a_full_table(B, O, monkeypatch):
    """The dense variant on every launch of a scan whose L7 table fills and is swept (continuous cleaning, tightening
    masks, lookup-only candidates, displacement chains), then 24 MiB with a 12 MiB copy (the long extent leaves through
    k_long_compare and comes back into the variant)."""
    monkeypatch.setenv("LRZGPU_RESOLVE_DENSE", "always")
    st = _check(B, O, datagen.text_like(24 * 1048576 + 5, seed=21), level=7)
    assert st.minimum_tag_mask > 1
    _check(B, O, datagen.long_range(16 * 1048576, seed=5, base_frac=0.5, mutate_every=70001), level=7)


def test_long_range_copy(B, O):
    """Half of the chunk is a copy of the other half at 12 MiB distance, with sparse mutations."""
    data = datagen.long_range(24 * 1048576, seed=5, base_frac=0.5, mutate_every=70001)
    st = _check(B, O, data, level=7)
    assert st.match_bytes > 11 * 1048576


@pytest.mark.parametrize("break_at", [None, 9 * 1048576 + 12345, 4 * 1048576 + 1, 12 * 1048576 - 3])
def test_long_match_extent_offload(B, O, break_at):
    """A 12 MiB exact copy: the resolver hands forward extents that are still equal after 4 MiB to the
    grid-wide compare (k_long_compare) and resumes; a single changed byte ends the match inside it."""
    half = 12 * 1048576
    base = bytearray(datagen.random_bytes(half, seed=31))
    data = bytearray(base + base)
    if break_at is not None:
        data[half + break_at] ^= 0x5A
    st = _check(B, O, bytes(data), level=7)
    assert st.match_bytes > 11 * 1048576


def test_per_tile_candidate_fallback(B, O, monkeypatch):
    """Segments with more candidates than the packed list holds fall back to the per-tile lists."""
    monkeypatch.setenv("LRZGPU_COMP_CAP", "1000")
    _check(B, O, datagen.text_like(3 * 1048576 + 5, seed=41), level=7)
    _check(B, O, datagen.long_range(2 * 1048576, seed=42), level=4)


def test_victim_round_carry(B, O):
    """Many identical tags force the round-robin victim path; the static victim_round is carried in/out."""
    pat = (b"0123456789abcdefghijklmnopqrstu" * 40 + b"XYZ") * 3000
    for vr in (0, 5):
        _check(B, O, pat, level=7, victim_round=vr)
        _check(B, O, pat, level=4, victim_round=vr % 3)


@pytest.mark.parametrize("mib", [4096])
def test_headline_shape(B, O, mib):
    """The bench workload shape (seeded text + identical copy at distance n/2), up to the full 4 GiB of
    BASELINE.json configs[1]: tag masks up to 0x1ff, clusters of ~340 slots, twins under continuous
    cleaning, and a multi-GiB match that goes through k_long_compare -- both streams and every statistic
    bit for bit against the oracle (its scan takes about a minute at 4 GiB)."""
    import importlib.util
    import os
    import torch
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("lrz_bench_w", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    n = mib << 20
    data = bench.make_cfg2(n, 1, torch.device("cuda:0"), "alnum")[:n].cpu().numpy().tobytes()
    st = _check(B, O, data, level=7)
    assert st.match_bytes > (n // 2) - (8 << 20) and st.minimum_tag_mask >= 0x3f
