"""The read side of the stream API (src/include/stream.h:20-21, 26, 29, 31, 32) through the C ABI, driven the way
runzip_fd() / runzip_chunk() / unzip_literal() / unzip_match() drive the reference's (src/runzip.c:139-440): read the
chunk_bytes byte, open_stream_in, token headers and match offsets from stream 0, literals from stream 1, write_1g of
what is rebuilt, the CRC after the terminator, close_stream_in, next chunk until eof, then the hash.  Images come
from the ORACLE (stored, reference-LZMA and zstd blocks, filters) -- no GPU involved."""
import ctypes as C
import hashlib
import os
import zlib

import pytest

import datagen
import test_filters_cpu as F

RAM = 80 * 100 * 1048576


def runzip(B, img: bytes, tmp_path, threads=4):
    """runzip_fd() in Python over lrzgpu_*: returns the reconstructed bytes."""
    L = B.lib()
    L.lrzgpu_open_stream_in.restype = C.c_void_p
    L.lrzgpu_open_stream_in.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_char]
    L.lrzgpu_read_stream.restype = C.c_int64
    L.lrzgpu_read_stream.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int64]
    L.lrzgpu_close_stream_in.argtypes = [C.c_void_p, C.c_void_p]
    L.lrzgpu_write_1g.restype = C.c_int64
    L.lrzgpu_write_1g.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
    L.lrzgpu_read_1g.restype = C.c_int64
    L.lrzgpu_read_1g.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int64]
    L.lrzgpu_get_readseek.restype = C.c_int64
    L.lrzgpu_get_readseek.argtypes = [C.c_void_p, C.c_int]
    m = B.read_magic(img)
    src, dst = tmp_path / "in.lrz", tmp_path / "out.bin"
    src.write_bytes(img)
    fi = os.open(src, os.O_RDONLY)
    fo = os.open(dst, os.O_RDWR | os.O_CREAT | os.O_TRUNC, 0o644)
    ctl = B.make_control(threads=threads, processors=threads, ramsize=RAM, filter_flag=m.filter_flag, delta=m.delta)
    ctl.fd_out = fo
    ctl.st_size = 0
    cp = C.byref(ctl)
    os.lseek(fi, m.magic_len + m.comment_length, os.SEEK_SET)

    def rd(ss, stream, n):
        buf = C.create_string_buffer(n)
        got = L.lrzgpu_read_stream(cp, ss, stream, buf, n)
        assert got == n, (stream, n, got)
        return buf.raw[:n]

    total = 0
    try:
        while True:
            cb = C.create_string_buffer(1)
            # "Reading chunk_bytes at ..." (src/runzip.c:293): the chunk header starts where the previous chunk ended
            assert L.lrzgpu_get_readseek(cp, fi) == os.lseek(fi, 0, os.SEEK_CUR) >= m.magic_len + m.comment_length
            assert L.lrzgpu_read_1g(cp, fi, cb, 1) == 1
            chunk_bytes = cb.raw[0]
            ss = L.lrzgpu_open_stream_in(cp, fi, 2, cb.raw[0:1])
            assert ss, "open_stream_in failed"
            start = total
            crc = 0
            while True:  # read_header(): u8 head, then the length in control->chunk_bytes = 2 bytes (src/runzip.c:315)
                head = rd(ss, 0, 1)[0]
                ln = int.from_bytes(rd(ss, 0, 2), "little")
                if head == 0 and ln == 0:
                    break
                if head == 0:  # unzip_literal
                    lit = rd(ss, 1, ln)
                    assert L.lrzgpu_write_1g(cp, lit, ln) == ln
                    crc = zlib.crc32(lit, crc)
                    total += ln
                else:  # unzip_match: offset back from the current output position, in chunk_bytes bytes
                    off = int.from_bytes(rd(ss, 0, chunk_bytes), "little")
                    assert 0 < off <= total - start
                    left = ln
                    while left:
                        k = min(left, off)
                        piece = os.pread(fo, k, total - off)
                        assert len(piece) == k
                        assert L.lrzgpu_write_1g(cp, piece, k) == k
                        crc = zlib.crc32(piece, crc)
                        total += k
                        left -= k
            want_crc = int.from_bytes(rd(ss, 0, 4), "big")  # read_u32 of the chunk trailer: stored most significant byte first
            assert want_crc == crc
            # the literal stream must be exhausted: one more byte is not there
            extra = C.create_string_buffer(1)
            assert L.lrzgpu_read_stream(cp, ss, 1, extra, 1) == 0
            assert L.lrzgpu_close_stream_in(cp, ss) == 0
            if ctl.eof:
                break
        # what follows the last chunk is the hash of the whole output (src/runzip.c:384-440)
        tail = os.read(fi, 1 << 16)
        assert len(tail) == m.hash_len
    finally:
        os.close(fi)
        os.close(fo)
    out = dst.read_bytes()
    assert len(out) == total
    if m.st_size:
        assert total == m.st_size
    return out, tail, ctl


@pytest.mark.parametrize("kind,kw", [("longrange", {}), ("text", {}), ("text", {"no_compress": 1}), ("random", {}), ("zeros", {}),
                                     ("longrange", {"zstd": 1, "zstd_level": 3})])
def test_replay_oracle_images_token_by_token(B, O, tmp_path, kind, kw):
    n = 3 * 1048576 + 1234
    data = datagen.KINDS[kind](n, seed=21)
    img, fs = O.compress_buffer(data, compression_level=7, threads=4, processors=8, workers=4, **kw)
    out, tail, ctl = runzip(B, img, tmp_path)
    assert out == data and tail == hashlib.md5(data).digest()


def test_replay_many_chunks_and_blocks(B, O, tmp_path):
    """three chunks of 20 MiB, 5 MiB blocks (small -m, -L1): the block chains of both streams, chunk after chunk"""
    data = datagen.long_range(45 * 1048576 + 77, seed=3)
    img, fs = O.compress_buffer(data, compression_level=1, threads=2, processors=2, ramsize=30 * 1048576, workers=4)
    assert fs.n_chunks == 3 and fs.n_blocks > 6
    out, tail, ctl = runzip(B, img, tmp_path, threads=3)
    assert out == data and tail == hashlib.md5(data).digest()
    assert ctl.st_size >= len(data)  # open_stream_in adds every chunk's size field


def test_replay_undoes_filters_and_handles_stdout_images(B, O, tmp_path):
    base = F.code_like(F.X86, 1048576 + 7, seed=4)
    data = base + datagen.text_like(300000, seed=4) + base[:400000]
    for flag, delta in ((F.X86, 0), (F.DELTA, 4)):
        img, _ = O.compress_buffer(data, filter_flag=flag, filter_delta=delta, compression_level=7, threads=4, processors=8, workers=4)
        out, tail, _ = runzip(B, img, tmp_path)
        assert out == data
    # an image as written to STDOUT from STDIN: several chunks, an empty last one, no size in the magic
    ram = 60 << 20
    chunk = (ram // 6) // 4096 * 4096
    data = datagen.long_range(2 * chunk, seed=6)
    img, fs = O.compress_buffer(data, compression_level=1, threads=2, processors=2, ramsize=ram, workers=4, stdin_mode=1, stdout_mode=1)
    assert fs.n_chunks == 3 and int.from_bytes(img[6:14], "little") == 0
    out, tail, _ = runzip(B, img, tmp_path)
    assert out == data and tail == hashlib.md5(data).digest()
    assert B.decompress_buffer(img, host_threads=2) == data  # the whole-file verifier copes with the size-less magic too


def test_stream_in_refuses_damaged_chunks(B, O, tmp_path):
    data = datagen.text_like(500000, seed=8)
    img, _ = O.compress_buffer(data, compression_level=7, threads=2, processors=2, workers=2)
    L = B.lib()
    L.lrzgpu_open_stream_in.restype = C.c_void_p
    L.lrzgpu_open_stream_in.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_char]
    bad = bytearray(img)
    bad[21 + 2 + img[21]] = 6  # the first initial stream header must be CTYPE_NONE
    p = tmp_path / "bad.lrz"
    p.write_bytes(bytes(bad))
    fi = os.open(p, os.O_RDONLY)
    try:
        os.lseek(fi, 22, os.SEEK_SET)
        ctl = B.make_control(threads=2, processors=2, ramsize=RAM)
        assert not L.lrzgpu_open_stream_in(C.byref(ctl), fi, 2, bytes([img[21]]))
        assert not L.lrzgpu_open_stream_in(C.byref(ctl), fi, 2, bytes([9]))
    finally:
        os.close(fi)
