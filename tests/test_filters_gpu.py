"""Filters on the device (src/stream.c:1587-1628): the kernels of csrc/filters_gpu.hip byte for byte against
the reference's own converters (oracle/_ref: Bra.c, Bra86.c, BraIA64.c, Delta.c compiled unmodified) and the library's
host converters; whole filtered .lrz IMAGES byte for byte against the oracle driver running the reference's converter
over every literal block; hash code and filter as per-run fields of the control, in the whole-file and the
chunk-sharded entry points."""
import ctypes as C
import hashlib
import random

import pytest

import datagen
import test_filters_cpu as F

pytestmark = pytest.mark.gpu
RAM = 80 * 100 * 1048576


@pytest.fixture(scope="module")
def R(O):
    r = O.ref_lzma()
    assert r is not None, "oracle/_ref/liblzma_ref.so missing"
    return r


def dev_filter(B, flag, delta, data):
    import torch
    L = B.lib()
    L.lrzgpu_filter_block_dev.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_int64, C.c_int]
    t = torch.frombuffer(bytearray(data) + bytearray(16), dtype=torch.uint8).cuda()
    rc = L.lrzgpu_filter_block_dev(flag, delta, t.data_ptr(), len(data), 0)
    assert rc == 0, rc
    out = bytes(t.cpu().numpy().tobytes())
    assert out[len(data):] == bytes(16)  # nothing written past the block
    return out[:len(data)]


@pytest.mark.parametrize("flag", [F.X86, F.ARM, F.ARMT, F.PPC, F.SPARC, F.IA64, F.ARM64, F.RISCV])
def test_device_filters_equal_reference(B, R, flag):
    for seed, n in enumerate([0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 15, 16, 17, 31, 33, 64, 100, 1001, 4099, 65537, 300003, 3 * 1048576 + 5]):
        for data in (F.code_like(flag, n, seed), datagen.KINDS["random"](n, seed=seed) if n else b""):
            want = F.ref_filter(R, flag, 0, data, True)
            assert dev_filter(B, flag, 0, data) == want, (flag, n)
            assert F.lib_filter(B, flag, 0, data, True) == want


def test_device_x86_histories_and_dense_fallback(B, R):
    """Bursts of opcode / sign bytes between stretches of filler (the candidate-list path with every history and skip
    case), and inputs of nothing but such bytes (every position a candidate: one long run)."""
    for seed in range(40):
        rnd = random.Random(seed)
        alphabet = rnd.choice([[0xE8, 0xE9, 0, 0xFF, 1], [0xE8, 0, 0xFF], [0xE8, 0xE9, 0, 0xFF, 0x7F, 0x80, 0xFE, 2], [0xE8, 0xFF]])
        out = bytearray()
        target = rnd.choice([200, 5000, 70000])
        while len(out) < target:
            out += bytes(rnd.choice(alphabet) for _ in range(rnd.randrange(1, 12)))
            out += bytes(rnd.choice([0x11, 0x00, 0xFF, 0x42]) for _ in range(rnd.choice([0, 1, 2, 5, 6, 7, 8, 9, 30, 90])))
        data = bytes(out)
        assert dev_filter(B, F.X86, 0, data) == F.ref_filter(R, F.X86, 0, data, True), seed
        dense = bytes(rnd.choice(alphabet) for _ in range(rnd.choice([5, 9, 64, 3000, 20000])))
        assert dev_filter(B, F.X86, 0, dense) == F.ref_filter(R, F.X86, 0, dense, True), seed


def test_device_riscv_dense_and_runs(B, R):
    import struct
    for seed in range(20):
        rnd = random.Random(100 + seed)
        n = rnd.choice([8, 10, 14, 64, 4096, 50000])
        out = bytearray(n)
        for i in range(0, n - 3, 4):  # nothing but JAL / AUIPC words: candidates in every parcel pair
            rd = rnd.choice([1, 5, 0, 2, 2, 7])
            op = rnd.choice([0x6F, 0x17, 0x17])
            hi = rnd.getrandbits(20) | (3 if rd == 2 else 0)
            struct.pack_into("<I", out, i, (hi << 12) | (rd << 7) | op)
        data = bytes(out)
        assert dev_filter(B, F.RISCV, 0, data) == F.ref_filter(R, F.RISCV, 0, data, True), seed


@pytest.mark.parametrize("delta", [1, 2, 3, 4, 16, 48, 256])
def test_device_delta_equals_reference(B, R, delta):
    for seed, n in enumerate([0, 1, delta - 1, delta, delta + 1, 1000, 65537, 2 * 1048576 + 3]):
        data = datagen.KINDS["text"](n, seed=seed) if n else b""
        assert dev_filter(B, F.DELTA, delta, data) == F.ref_filter(R, F.DELTA, delta, data, True), (delta, n)


@pytest.mark.parametrize("flag,delta", [(F.X86, 0), (F.ARM64, 0), (F.RISCV, 0), (F.ARMT, 0), (F.IA64, 0), (F.DELTA, 4), (F.DELTA, 48)])
def test_filtered_image_equals_oracle_image(B, O, flag, delta):
    """The whole .lrz with a filter selected in the control: byte-identical to the oracle driver with the REFERENCE's
    converter over every literal block (compressed and stored blocks, -n too), magic[16] set, lz4 test off; decodes."""
    base = F.code_like(flag if flag != F.DELTA else F.ARM, 1048576 + 7, seed=flag + delta)
    data = base + datagen.text_like(300000, seed=flag) + base[:500000]  # literals, text, a long-range repeat
    for kw, okw in (({}, {}), ({"no_compress": True}, {"no_compress": 1})):
        want, _ = O.compress_buffer(data, filter_flag=flag, filter_delta=delta, compression_level=7, threads=4, processors=8,
                                    workers=4, **okw)
        got, ctl = B.compress_buffer(data, level=7, threads=4, processors=8, host_threads=4, filter_flag=flag, delta=delta, **kw)
        assert got[16] == O.magic16(flag, delta) and got[14] == 1
        assert got == want, (flag, delta, kw)
        assert B.decompress_buffer(got, host_threads=2) == data
    # no filter in the control: the plain image, whatever ran before in this process
    want, _ = O.compress_buffer(data, compression_level=7, threads=4, processors=8, workers=4)
    got, _ = B.compress_buffer(data, level=7, threads=4, processors=8, host_threads=4)
    assert got == want and got[16] == 0


def salted_noise(n, seed):
    """noise with CALL-like byte patterns (E8/E9 xx xx xx 00/FF) every ~40 bytes"""
    import numpy as np
    rng = np.random.default_rng(seed)
    x = rng.integers(0, 256, n, dtype=np.uint8)
    idx = rng.choice(n - 8, n // 40, replace=False)
    x[idx] = np.where(rng.random(idx.size) < 0.8, 0xE8, 0xE9)
    x[idx + 4] = np.where(rng.random(idx.size) < 0.5, 0x00, 0xFF)
    return x.tobytes()


def test_filter_with_small_blocks_and_many_chunks(B, O):
    """Several chunks and several literal blocks per chunk (a small -m: 20 MiB chunks, 5 MiB blocks), x86 filter over
    noise salted with CALL-like bytes."""
    data = salted_noise(45 * 1048576 + 321, 7)
    kw = dict(threads=2, processors=2, ramsize=30 * 1048576)
    want, fs = O.compress_buffer(data, filter_flag=F.X86, compression_level=1, workers=4, **kw)
    got, _ = B.compress_buffer(data, level=1, host_threads=4, filter_flag=F.X86, **kw)
    assert fs.n_chunks == 3 and fs.n_blocks > 12
    assert got == want
    assert B.decompress_buffer(got, host_threads=2) == data


def test_hash_code_is_a_field_of_the_control(B, O):
    data = datagen.long_range(2 * 1048576 + 99, seed=77)
    want, _ = O.compress_buffer(data, compression_level=7, threads=4, processors=8, workers=4)
    got, ctl = B.compress_buffer(data, level=7, threads=4, processors=8, host_threads=4)
    assert got == want and bytes(ctl.hash_resblock) == hashlib.md5(data).digest()
    for code, ref in ((3, hashlib.sha256(data).digest()), (5, hashlib.sha512(data).digest()), (6, hashlib.sha3_256(data).digest()),
                      (9, hashlib.shake_128(data).digest(32)), (0, b"")):
        img, ctl = B.compress_buffer(data, level=7, threads=4, processors=8, host_threads=4, hash_code=code)
        assert img[14] == code and len(img) == len(want) - 16 + len(ref)
        assert img[:14] == want[:14] and img[15:len(want) - 16] == want[15:len(want) - 16]
        assert img[len(want) - 16:] == ref and bytes(ctl.hash_full)[:len(ref)] == ref
        assert B.decompress_buffer(img, host_threads=2) == data
        # the next run of the same process with a plain control is back at MD5: nothing global was touched
        again, _ = B.compress_buffer(data, level=7, threads=4, processors=8, host_threads=4)
        assert again == want
    with pytest.raises(RuntimeError):
        B.compress_buffer(data, level=7, threads=4, processors=8, hash_code=14)
    with pytest.raises(RuntimeError):
        B.compress_buffer(data, level=7, threads=4, processors=8, filter_flag=F.DELTA, delta=20)


def test_sharded_entry_points_honour_hash_and_filter(B):
    """lrzgpu_compress_chunks + lrzgpu_assemble_chunks with a SHA-256 and the x86 filter in the control: the same
    file as the whole-file entry point."""
    data = salted_noise(45 * 1048576 + 99, 8)
    kw = dict(level=1, threads=2, processors=2, ramsize=30 * 1048576, host_threads=4, hash_code=3, filter_flag=F.X86)
    whole, ctl = B.compress_buffer(data, **kw)
    _, chunk = B.plan(len(data), **{k: v for k, v in kw.items() if k != "host_threads"})
    n_chunks = (len(data) + chunk - 1) // chunk
    assert n_chunks == 3
    imgs = {}
    digest = None
    for first in range(2):
        got, c = B.compress_chunks(data=data, first=first, stride=2, with_md5=(first == 0), ctl=B.make_control(**kw))
        for k, (vin, vout, img) in got.items():
            imgs[k] = img
        if first == 0:
            digest = bytes(c.hash_full)[:32]
    assert digest == hashlib.sha256(data).digest()
    out, _ = B.assemble_chunks([imgs[k] for k in range(n_chunks)], len(data), digest, ctl=B.make_control(**kw))
    assert bytes(out) == whole and whole[14] == 3 and whole[16] == 1
    assert B.decompress_buffer(whole, host_threads=2) == data
