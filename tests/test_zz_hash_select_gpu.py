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
