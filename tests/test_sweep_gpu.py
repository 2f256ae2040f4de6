"""Seeded random sweep of the whole path against the oracle: levels, -p, -w, lz4 gate on/off,
threshold, input kind and size drawn at random (fixed seeds, so every run checks the same cases)."""
import random

import pytest

import datagen

pytestmark = pytest.mark.gpu
RAM = 80 * 100 << 20


def _case(seed):
    r = random.Random(seed)
    kind = r.choice(["text", "random", "phrases", "sparse", "longrange", "few", "zeros"])
    n = r.choice([0, 1, 30, 31, 32, 4095, 65537, r.randrange(100000, 3000000), r.randrange(3000000, 12000000)])
    if kind in ("few", "zeros", "phrases"):
        n = min(n, 1500000)  # collapsed tag spaces run in serial resolver steps: keep them small
    kw = dict(level=r.randrange(1, 10), threads=r.choice([1, 2, 3, 4, 8, 16]), processors=r.choice([1, 4, 8, 16]),
              lz4_test=r.random() < 0.8, threshold=r.choice([100, 100, 95, 80, 50]))
    if r.random() < 0.15:
        kw["no_compress"] = True
    elif r.random() < 0.15:
        kw["zstd"] = True
        kw["zstd_level"] = r.choice([0, 0, 1, 5, 12, 19, 22])
    return kind, n, kw


# One seed per distinct (input kind, size class (tiny / up to 3 MB / larger), back end: fast / optimal LZMA, stored, zstd) among seeds 1000..1159 --
# the 55 of 160 that do not repeat a combination an earlier seed already covers (the GPU suite has a time budget).
SEEDS = [1000, 1001, 1002, 1003, 1004, 1005, 1006, 1007, 1009, 1010, 1015, 1016, 1019, 1020, 1021, 1024, 1025, 1026,
         1029, 1030, 1031, 1033, 1034, 1037, 1039, 1040, 1041, 1043, 1045, 1048, 1049, 1051, 1052, 1053, 1054, 1058,
         1060, 1062, 1063, 1064, 1066, 1069, 1070, 1072, 1073, 1082, 1085, 1088, 1103, 1109, 1111, 1135, 1143, 1147,
         1154]


@pytest.mark.parametrize("seed", SEEDS)
def test_random_parameters(B, O, seed):
    kind, n, kw = _case(seed)
    data = datagen.KINDS[kind](n, seed=seed)
    okw = dict(compression_level=kw["level"], threads=kw["threads"], processors=kw["processors"], ramsize=RAM,
               no_compress=int(kw.get("no_compress", False)), lz4_test=int(kw["lz4_test"]), threshold=kw["threshold"], workers=8,
               zstd=int(kw.get("zstd", False)), zstd_level=kw.get("zstd_level", 0))
    want, fs = O.compress_buffer(data, **okw)
    got, ctl = B.compress_buffer(data, ramsize=RAM, host_threads=8, **kw)
    assert ctl.stream_bufsize == fs.stream_bufsize and ctl.dictSize_used == fs.dict_size, (kind, n, kw)
    assert got == want, (kind, n, kw, len(got), len(want))
    assert B.decompress_buffer(got, host_threads=4) == data
