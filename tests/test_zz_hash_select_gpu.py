"""The whole-file hash of the compress path (lrzgpu_select_hash == -H of the reference's command line): the default
stays MD5 and byte-identical to the oracle; any other code of hashes[] lands in magic[14] with its digest after the
last chunk, and the read side verifies it.  (Named to run last: it changes a process-wide setting.)"""
import ctypes as C
import hashlib

import pytest

import datagen

pytestmark = pytest.mark.gpu
RAM = 80 * 100 * 1048576


def test_compress_with_selected_hash(B, O):
    L = B.lib()
    data = datagen.long_range(2 * 1048576 + 99, seed=77)
    want, _ = O.compress_buffer(data, compression_level=7, threads=4, processors=8, workers=4)
    try:
        got, ctl = B.compress_buffer(data, level=7, threads=4, processors=8, host_threads=4)
        assert got == want and bytes(ctl.hash_resblock) == hashlib.md5(data).digest()
        for code, ref, n in ((3, hashlib.sha256(data).digest(), 32), (5, hashlib.sha512(data).digest(), 64),
                             (6, hashlib.sha3_256(data).digest(), 32), (9, hashlib.shake_128(data).digest(32), 32), (0, b"", 0)):
            assert L.lrzgpu_select_hash(code) == 0
            img, ctl = B.compress_buffer(data, level=7, threads=4, processors=8, host_threads=4)
            assert img[14] == code and len(img) == len(want) - 16 + n
            assert img[:14] == want[:14] and img[15:len(want) - 16] == want[15:len(want) - 16]
            assert img[len(want) - 16:] == ref
            assert B.decompress_buffer(img, host_threads=2) == data
        assert L.lrzgpu_select_hash(14) != 0
    finally:
        assert L.lrzgpu_select_hash(1) == 0
    got, _ = B.compress_buffer(data, level=7, threads=4, processors=8, host_threads=4)
    assert got == want


def test_compress_with_selected_filter(B, O):
    """lrzgpu_select_filter == --x86 / --arm64 / --delta=N: every literal block is filtered before its back end (host
    converters, pinned to the reference's in tests/test_filters_cpu.py), magic[16] names the filter, the lz4 test is off,
    and the read side gives the input back; the default (no filter) stays byte-identical to the oracle."""
    import test_filters_cpu as F
    L = B.lib()
    L.lrzgpu_select_filter.argtypes = [C.c_int, C.c_int]
    want_by_flag = {F.X86: 1, F.ARM64: 7, F.RISCV: 8}
    try:
        for flag, delta, magic16 in ((F.X86, 0, 1), (F.ARM64, 0, 7), (F.RISCV, 0, 8), (F.DELTA, 4, 132), (F.DELTA, 48, 146)):
            base = F.code_like(flag if flag != F.DELTA else F.ARM, 1048576 + 7, seed=flag + delta)
            data = base + datagen.text_like(300000, seed=flag) + base[:500000]  # literals, text, a long-range repeat
            assert L.lrzgpu_select_filter(flag, delta) == 0
            for kw in ({}, {"no_compress": True}):
                img, ctl = B.compress_buffer(data, level=7, threads=4, processors=8, host_threads=4, **kw)
                assert img[16] == magic16 and img[14] == 1
                assert B.decompress_buffer(img, host_threads=2) == data
        assert L.lrzgpu_select_filter(F.DELTA, 20) != 0 and L.lrzgpu_select_filter(9, 0) != 0
    finally:
        assert L.lrzgpu_select_filter(0, 0) == 0
    data = datagen.long_range(2 * 1048576 + 99, seed=78)
    want, _ = O.compress_buffer(data, compression_level=7, threads=4, processors=8, workers=4)
    got, _ = B.compress_buffer(data, level=7, threads=4, processors=8, host_threads=4)
    assert got == want and got[16] == 0
