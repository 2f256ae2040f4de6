"""The duplicate census of the rzip scan (csrc/rzip_census.hip): "no 31-byte window of this chunk occurs twice", answered
exactly -- a match of the scan is at least MINIMUM_MATCH = 31 bytes long (src/rzip.c:431-461), so such a chunk (the
file's last: nobody needs the victim_round it ends with) is not put through the table automaton at all.  The verdict
against brute force, and whole images against the oracle's with the census on and off."""
import ctypes as C

import numpy as np
import pytest

import datagen
from test_compress_gpu import _both

pytestmark = pytest.mark.gpu


def _census(B, data):
    f = B.lib().lrzgpu_census
    f.argtypes = [C.c_char_p, C.c_int64, C.c_int, C.POINTER(C.c_int64)]
    st = (C.c_int64 * 5)()
    v = f(bytes(data), len(data), 0, st)
    assert v in (0, 1), v
    return v, list(st)


def _has_repeat(data):
    seen = set()
    for i in range(len(data) - 30):
        w = data[i:i + 31]
        if w in seen:
            return True
        seen.add(w)
    return False


def test_verdict_on_plain_cases(B):
    rnd = datagen.random_bytes((8 << 20) + 13, seed=11)
    v, st = _census(B, rnd)
    assert v == 1 and st[1] == 0 and st[3] == 0
    # ~2 of 25 positions are anchors, a 64th of them in the sample
    assert 0.06 * len(rnd) < st[2] < 0.10 * len(rnd) and st[2] / 80 < st[0] < st[2] / 50, st
    assert _census(B, b"")[0] == 1 and _census(B, rnd[:30])[0] == 1  # no 31-byte window at all
    assert _census(B, rnd[:31] + rnd[:31])[0] == 0
    assert _census(B, bytes(1 << 20))[0] == 0                       # every position an anchor: no room for them, "maybe"
    assert _census(B, datagen.text_like(4 << 20, seed=3))[0] == 0    # (the sample says so)


def test_equal_values_that_are_chance_are_cleared(B):
    """Two anchors with the same 8 bytes whose surroundings differ (eight zero bytes in noise: the smallest value there is,
    an anchor for certain): the sort finds the pair, the look at the bytes around the two positions clears it -- and with
    30 equal bytes around them still, with 31 no more."""
    rnd = bytearray(datagen.random_bytes((6 << 20) + 5, seed=77))
    d = bytearray(rnd)
    d[1000003:1000011] = bytes(8)
    d[5000017:5000025] = bytes(8)
    v, st = _census(B, d)
    assert v == 1 and st[3] >= 1 and st[4] == st[3], st
    for before, total, want in ((0, 30, 1), (11, 30, 1), (22, 30, 1), (0, 31, 0), (11, 31, 0), (23, 31, 0), (5, 40, 0)):
        e = bytearray(d)
        e[5000017 - before:5000017 - before + total] = e[1000003 - before:1000003 - before + total]
        assert _census(B, e)[0] == want, (before, total)


@pytest.mark.parametrize("gap", [0, 1, 100, 3 << 20])
@pytest.mark.parametrize("length", [31, 32, 64, 1000])
def test_one_repeat_is_seen(B, gap, length):
    """Noise with ONE stretch of `length` bytes copied `gap` bytes behind itself -- at every alignment the copy has to be
    seen: all its 31-byte windows are repeats, and each holds an anchor at the same relative place as its original."""
    rnd = bytearray(datagen.random_bytes((5 << 20) + 7, seed=100 + length))
    for src in (12345, 1 << 20, (1 << 20) + 3):
        d = bytearray(rnd)
        dst = src + length + gap
        d[dst:dst + length] = d[src:src + length]
        assert _census(B, d)[0] == 0, (src, gap, length)


@pytest.mark.parametrize("seed", range(6))
def test_verdict_one_is_true(B, seed):
    """Small inputs over small alphabets (repeats likely, ties between equal 8-byte values everywhere): whenever the
    census says "none", brute force must agree."""
    rng = np.random.default_rng(seed)
    said_none = 0
    for trial in range(40):
        n = int(rng.integers(31, 3000))
        k = int(rng.choice([2, 3, 4, 16, 256]))
        d = rng.integers(0, k, size=n, dtype=np.uint8).tobytes()
        v, _ = _census(B, d)
        if v == 1:
            said_none += 1
            assert not _has_repeat(d), (seed, trial, n, k)
    assert said_none >= 5


@pytest.mark.parametrize("census", ["1", "0"])
def test_images_equal_the_oracle_with_and_without(B, O, monkeypatch, census):
    """Whole .lrz images of incompressible input, of noise with a single repeat, and of a last chunk of noise behind
    chunks of text, with the census allowed and with every chunk through the resolver."""
    monkeypatch.setenv("LRZGPU_CENSUS", census)
    rnd = datagen.random_bytes((34 << 20) + 11, seed=21)
    _both(B, O, rnd, level=7, threads=16, processors=16)
    d = bytearray(rnd[:24 << 20])
    d[20 << 20:(20 << 20) + 5000] = d[1 << 20:(1 << 20) + 5000]
    _both(B, O, bytes(d), level=7, threads=16, processors=16)
    mixed = datagen.text_like(100 << 20, seed=5) + rnd[:20 << 20]  # -w 1: the second, last chunk is the noise
    fs = _both(B, O, mixed, level=3, threads=4, processors=8, window=1)  # (level 3: the oracle's LZMA of 100 MiB of text in seconds)
    assert fs.n_chunks == 2
