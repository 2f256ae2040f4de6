"""CPU tests: the oracle against golden vectors (recorded reference outputs + committed fixtures)."""
import hashlib
import json
import os
import zlib

import pytest

import datagen

HERE = os.path.dirname(os.path.abspath(__file__))
KA = json.load(open(os.path.join(HERE, "golden", "known_answers.json")))
sha = lambda b: hashlib.sha256(b).hexdigest()


@pytest.fixture(scope="module")
def syn64():
    import gen_syn
    d = gen_syn.syn64()
    rec = KA["reference_recorded"]["input"]
    assert len(d) == rec["size"] and sha(d) == rec["sha256"]
    assert hashlib.md5(d).hexdigest() == rec["md5"]
    return d


def test_hash_index_matches_recorded_reference_values(O):
    hx = O.hash_index()
    assert hx[0] == KA["reference_recorded"]["hash_index_0"]
    assert hx[1] == KA["reference_recorded"]["hash_index_1"]
    assert hx == KA["hash_index"]
    assert all(0 <= v < (1 << 47) for v in hx)


def test_k2_rzip_only_equals_reference_binary_output(O, syn64):
    """-n -p1 -m80 on syn64: byte-identical (sha256) to what the reference binary wrote."""
    rec = KA["reference_recorded"]["k2"]
    out, fs = O.compress_buffer(syn64, no_compress=1, threads=1, ramsize=80 * 100 * 1048576)
    assert len(out) == rec["size"] and sha(out) == rec["sha256"]
    assert fs.rz.inserts == rec["inserts"] and fs.rz.tag_hits == rec["true_tag_positives"]
    assert fs.rz.tag_misses == rec["false_tag_positives"]
    assert fs.rz.matches == rec["matches"] and fs.rz.literals == rec["literals"]
    # trailer: MD5 of the input
    assert out[-16:] == hashlib.md5(syn64).digest()


def test_k1_full_path_equals_reference_binary_output(O, syn64):
    """-L7 -p4 -m80 on syn64 (rzip + lz4 gate + reference LZMA + container): recorded sha256."""
    if O.ref_lzma() is None:
        pytest.skip("oracle/_ref/liblzma_ref.so not present")
    rec = KA["reference_recorded"]["k1"]
    out, fs = O.compress_buffer(syn64, compression_level=7, threads=4, processors=8, ramsize=80 * 100 * 1048576, workers=2)
    assert fs.stream_bufsize == rec["stream_bufsize"] and fs.threads_used == rec["threads"]
    assert len(out) == rec["size"] and sha(out) == rec["sha256"]


def test_k3_reference_defaults_equal_reference_binary_output(O, syn64):
    """The reference binary run with its DEFAULTS on the survey container (third row: 8 cores => 9 compress
    slots, 62 GiB of RAM): the other branch of open_stream_out's stream_bufsize rule (src/stream.c:1311-1323: limit /
    threads is not above 10 MiB => 10 MiB blocks) and seven LZMA blocks in one chunk; recorded sha256."""
    if O.ref_lzma() is None:
        pytest.skip("oracle/_ref/liblzma_ref.so not present")
    rec = KA["reference_recorded"]["k3"]
    out, fs = O.compress_buffer(syn64, compression_level=7, threads=8, processors=rec["processors"], ramsize=rec["ramsize"], workers=4)
    assert fs.stream_bufsize == rec["stream_bufsize"] and fs.threads_used == rec["threads"]
    assert len(out) == rec["size"] and sha(out) == rec["sha256"]
    k1 = KA["reference_recorded"]["k1"]
    assert (rec["size"], rec["sha256"]) != (k1["size"], k1["sha256"])


@pytest.mark.parametrize("case", KA["oracle"], ids=lambda c: "%s-%d-L%d" % (c["kind"], c["n"], c["level"]))
def test_oracle_stream_fixtures(O, case):
    data = datagen.KINDS[case["kind"]](case["n"], seed=case["seed"])
    s0, s1, st, crc, vr = O.rzip_chunk(data, level=case["level"])
    assert (len(s0), sha(s0)) == (case["stream0_len"], case["stream0_sha256"])
    assert (len(s1), sha(s1)) == (case["stream1_len"], case["stream1_sha256"])
    assert crc == case["crc32"] == (zlib.crc32(data) & 0xFFFFFFFF)
    assert st.matches == case["matches"] and st.inserts == case["inserts"] and vr == case["victim_round"]
    # stream 0 always ends with terminator + big-endian CRC
    assert s0[-7:-4] == b"\x00\x00\x00" and s0[-4:] == crc.to_bytes(4, "big")


def test_rzip_streams_round_trip(O):
    """Decode stream0/stream1 (runzip semantics, src/runzip.c:139-370) and get the input back."""
    for kind, n in (("longrange", 2000000), ("phrases", 300000), ("text", 100000)):
        data = datagen.KINDS[kind](n, seed=3)
        s0, s1, st, crc, vr = O.rzip_chunk(data, level=7)
        cb = 3 if n < (1 << 24) else 4
        out = bytearray()
        i = j = 0
        while True:
            head, ln = s0[i], s0[i + 1] | (s0[i + 2] << 8)
            i += 3
            if head == 0:
                if ln == 0:
                    break
                out += s1[j:j + ln]
                j += ln
            else:
                dist = int.from_bytes(s0[i:i + cb], "little")
                i += cb
                for _ in range(ln):  # byte-wise: source may overlap
                    out.append(out[-dist])
        assert bytes(out) == data and j == len(s1)


@pytest.mark.parametrize("case", KA["ref_lzma"], ids=lambda c: "%s-%d" % (c["kind"], c["n"]))
def test_ref_lzma_fixtures_and_mf_oracle(O, B, case):
    """The committed LzmaCompress hashes come from oracle/_ref; the product's host parser fed with
    the ORACLE match lists must reproduce them (CPU-only check of the host half)."""
    import numpy as np
    data = datagen.KINDS[case["kind"]](case["n"], seed=case["seed"])
    if O.ref_lzma() is not None:
        rc, comp, props = O.lzma_compress_ref(data, level=case["level"], dict_size=case["dict"], fb=case["fb"])
        assert rc == case["rc"] and sha(comp) == case["sha256"] and props.hex() == case["props"]
    if case["level"] < 5:  # algo 0: hash chains, cut value halved (LzmaEnc.c:95-99)
        offs, pairs = O.mf_hc5(data, dict_size=case["dict"], fb=case["fb"], cut=(16 + case["fb"] // 2) >> 1)
    else:
        offs, pairs = O.mf_bt4(data, dict_size=case["dict"], fb=case["fb"], cut=16 + case["fb"] // 2)
    counts = np.diff(offs).astype(np.uint8)
    rc, mine = B.lzma_encode_with_lists(data, counts, pairs, level=case["level"], dict_size=case["dict"], fb=case["fb"])
    assert rc == case["rc"] and len(mine) == case["size"] and sha(mine) == case["sha256"]


def test_lz4_oracle_against_system_liblz4(O):
    import ctypes as C
    try:
        lz4 = C.CDLL("liblz4.so.1")
    except OSError:
        pytest.skip("liblz4.so.1 not installed")
    lz4.LZ4_compress_default.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int]
    for kind in ("text", "random", "few", "phrases", "sparse", "zeros"):
        for n in (0, 1, 12, 13, 64, 1000, 65546, 65547, 200000, 1 << 20):
            data = datagen.KINDS[kind](n, seed=n % 13 + 1)
            for cap in (n + 1, n, n // 2, n + n // 255 + 16):
                dst = C.create_string_buffer(max(cap, 1))
                assert O.lib().lrzo_lz4_compress_default_size(data, n, cap) == lz4.LZ4_compress_default(data, dst, n, cap)


def test_lz4_early_verdict_bound(O):
    """The gate's early exit claims: once out_so_far + rest + rest/255 + 16 < bound, the final LZ4 size
    is below the bound too.  Checked on the CPU restatement over many inputs, sizes and bounds: an
    early stop always returns a value >= the exact size and < the bound, and no early stop means the
    exact size."""
    import ctypes as C
    import random
    L = O.lib()
    L.lrzo_lz4_size_stop_below.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int)]
    r = random.Random(7)
    early = 0
    for trial in range(600):
        kind = r.choice(["text", "random", "few", "phrases", "sparse", "zeros", "longrange"])
        n = r.choice([13, 64, 1000, 65546, 65547, r.randrange(2000, 300000)])
        data = datagen.KINDS[kind](n, seed=trial)
        if r.random() < 0.3:  # mixed: a compressible head and an incompressible tail, and the reverse
            other = datagen.random_bytes(n, seed=trial + 1)
            cut = r.randrange(n + 1)
            data = (data[:cut] + other[cut:]) if r.random() < 0.5 else (other[:cut] + data[cut:])
        exact = L.lrzo_lz4_compress_default_size(data, n, n + 1)
        bound = max(1, int(n * r.choice([1.0, 1.0, 0.95, 0.8, 0.5, 0.2])))
        flag = C.c_int(0)
        got = L.lrzo_lz4_size_stop_below(data, n, n + 1, bound, C.byref(flag))
        if flag.value:
            early += 1
            assert 0 < exact <= got < bound, (kind, n, bound, exact, got)
        else:
            assert got == exact, (kind, n, bound, exact, got)
    assert early > 100


@pytest.mark.parametrize("kind", ["text", "few", "phrases", "zeros"])
def test_bt4_lists_are_prefix_computable(O, kind):
    """What the early start of a block rests on: the BT4 lists of the positions below
    P - fb depend only on bytes [0, P) -- a bucket's walk visits earlier positions only and compares at most fb bytes.
    Checked with the restated reference finder: the lists of a prefix equal the lists of the whole block up to there
    (same dictionary, no larger than the prefix, so that both runs derive the same hash mask)."""
    import numpy as np
    import datagen
    n, fb, dict_size = 1500000, 64, 1 << 19
    data = datagen.KINDS[kind](n, seed=77)
    full_offs, full_pairs = O.mf_bt4(data, dict_size=dict_size, fb=fb, cut=16 + fb // 2)
    for P in (600000, 1048576 + 13):
        offs, pairs = O.mf_bt4(data[:P], dict_size=dict_size, fb=fb, cut=16 + fb // 2)
        safe = P - fb  # positions whose fb bytes of look-ahead lie inside the prefix
        assert np.array_equal(offs[:safe + 1], full_offs[:safe + 1])
        assert np.array_equal(pairs[:int(offs[safe])], full_pairs[:int(full_offs[safe])])
        # and the clipping is real: the prefix run's last positions see a shorter look-ahead
        if kind == "zeros":
            assert not np.array_equal(pairs[int(offs[P - 8]):int(offs[P - 7])], full_pairs[int(full_offs[P - 8]):int(full_offs[P - 7])])
