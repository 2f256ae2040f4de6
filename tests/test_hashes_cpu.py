"""Container / front-end completeness, host only: the whole-file hashes a .lrz may carry
(`hashes[]`, src/main.c:64-79) against Python's hashlib and published test vectors, the read side verifying every
one of them (src/runzip.c:352-440), the trailer rewrite, and read_magic()/get_magic() for archive versions 0.6-0.14
(src/lrzip.c:262-585, doc/magic.header.txt)."""
import ctypes as C
import hashlib
import random
import zlib

import pytest

import datagen

RAM = 80 * 100 * 1048576
LENGTHS = [4, 16, 20, 32, 48, 64, 32, 64, 16, 32, 64, 16, 32, 64]
LABELS = ["CRC", "MD5", "RIPEMD", "SHA256", "SHA384", "SHA512", "SHA3_256", "SHA3_512", "SHAKE128_16", "SHAKE128_32",
          "SHAKE128_64", "SHAKE256_16", "SHAKE256_32", "SHAKE256_64"]


class Magic(C.Structure):
    _fields_ = [("major", C.c_int), ("minor", C.c_int), ("magic_len", C.c_int), ("st_size", C.c_int64), ("enc_code", C.c_int),
                ("salt", C.c_uint8 * 8), ("costfactor", C.c_int), ("hash_code", C.c_int), ("hash_len", C.c_int),
                ("filter_flag", C.c_int), ("delta", C.c_int), ("ctype", C.c_int), ("dict_size", C.c_uint32),
                ("lzma_properties", C.c_uint8 * 5), ("zpaq_bs", C.c_int), ("zpaq_level", C.c_int), ("bzip3_bs", C.c_int),
                ("zstd_strategy", C.c_int), ("zstd_level", C.c_int), ("level", C.c_int), ("rzip_level", C.c_int),
                ("comment_length", C.c_int), ("comment", C.c_char * 256)]


@pytest.fixture(scope="module")
def L(B):
    lib = B.lib()
    lib.lrzgpu_hash_length.restype = C.c_int
    lib.lrzgpu_hash_label.restype = C.c_char_p
    lib.lrzgpu_hash_buffer.argtypes = [C.c_int, C.c_char_p, C.c_int64, C.c_char_p]
    lib.lrzgpu_hash_open.restype = C.c_void_p
    lib.lrzgpu_hash_update.argtypes = [C.c_void_p, C.c_char_p, C.c_int64]
    lib.lrzgpu_hash_final.argtypes = [C.c_void_p, C.c_char_p]
    lib.lrzgpu_set_file_hash.argtypes = [C.c_char_p, C.c_int64, C.c_int, C.c_char_p, C.POINTER(C.POINTER(C.c_ubyte)), C.POINTER(C.c_int64)]
    lib.lrzgpu_read_magic.argtypes = [C.c_char_p, C.c_int64, C.POINTER(Magic)]
    return lib


def reference_digest(code, data):
    """An independent implementation of the same function, or None where Python has none (RIPEMD-160 without OpenSSL legacy)."""
    if code == 0:
        return zlib.crc32(data).to_bytes(4, "big")  # libgcrypt hands CRC32 back most significant byte first
    if code == 2:
        try:
            return hashlib.new("ripemd160", data).digest()
        except (ValueError, TypeError):
            return None
    if code >= 8:
        h = hashlib.shake_128(data) if code <= 10 else hashlib.shake_256(data)
        return h.digest(LENGTHS[code])
    return {1: hashlib.md5, 3: hashlib.sha256, 4: hashlib.sha384, 5: hashlib.sha512, 6: hashlib.sha3_256, 7: hashlib.sha3_512}[code](data).digest()


def digest(L, code, data):
    out = C.create_string_buffer(64)
    assert L.lrzgpu_hash_buffer(code, data, len(data), out) == 0
    return out.raw[:LENGTHS[code]]


@pytest.mark.parametrize("code", range(14))
def test_hash_equals_hashlib(L, code):
    assert L.lrzgpu_hash_length(code) == LENGTHS[code] and L.lrzgpu_hash_label(code).decode() == LABELS[code]
    rnd = random.Random(code)
    # around every block / rate boundary of the functions involved (64, 128, 72, 136, 168 bytes and their padding limits)
    for n in [0, 1, 3, 55, 56, 57, 63, 64, 65, 71, 72, 73, 111, 112, 113, 119, 120, 127, 128, 129, 135, 136, 137, 143, 144, 145,
              167, 168, 169, 335, 336, 337, 1000, 4096, 100003]:
        data = bytes(rnd.getrandbits(8) for _ in range(n))
        want = reference_digest(code, data)
        if want is not None:
            assert digest(L, code, data) == want, (code, n)
    # streaming in ragged pieces == one shot
    data = datagen.text_like(300000 + code, seed=code)
    h = L.lrzgpu_hash_open(code)
    assert h
    at = 0
    while at < len(data):
        k = rnd.choice([1, 7, 63, 64, 65, 200, 4096, 50000])
        assert L.lrzgpu_hash_update(h, data[at:at + k], len(data[at:at + k])) == 0
        at += k
    out = C.create_string_buffer(64)
    assert L.lrzgpu_hash_final(h, out) == 0
    assert out.raw[:LENGTHS[code]] == digest(L, code, data)


def test_ripemd160_published_vectors(L):
    """RIPEMD-160 test vectors of the algorithm's authors (Dobbertin, Bosselaers, Preneel)."""
    for msg, want in ((b"", "9c1185a5c5e9fc54612808977ee8f548b2258d31"), (b"a", "0bdc9d2d256b3ee9daae347be6f4dc835a467ffe"),
                      (b"abc", "8eb208f7e05d987a9b044a8e98c6b087f15a0bfc"), (b"message digest", "5d0689ef49d2fae572b881b123a85ffa21595f36"),
                      (b"abcdefghijklmnopqrstuvwxyz", "f71c27109c692c1b56bbdceb5b9d2865b3708dbc"),
                      (b"abcdbcdecdefdefgefghfghighijhijkijkljklmklmnlmnomnopnopq", "12a053384a9c0c88e405a06c27dcf49ada62eb2b"),
                      (b"ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789", "b0e20b6e3116640286ed3a87a5713079b21f5189"),
                      (b"1234567890" * 8, "9b752e45573d4b39f4dbd3323cab82bf63326bfb"), (b"a" * 1000000, "52783243c1697bdbe16d37f97f68f08325dc1528")):
        assert digest(L, 2, msg).hex() == want
    assert L.lrzgpu_hash_length(14) == -1 and L.lrzgpu_hash_open(14) is None and L.lrzgpu_hash_open(-1) is None


@pytest.mark.parametrize("code", range(14))
def test_read_side_verifies_every_hash(B, O, L, code):
    """An image with hash code `code` (magic[14], digest after the last chunk) decodes, and a digest that is off
    by one bit is refused -- for LZMA blocks written by the oracle and for a multi-chunk stored image."""
    data = datagen.long_range(1048576 + 777, seed=40 + code)
    img, _ = O.compress_buffer(data, compression_level=7, threads=4, processors=4, ramsize=RAM)
    out, olen = C.POINTER(C.c_ubyte)(), C.c_int64()
    dg = digest(L, code, data)
    assert L.lrzgpu_set_file_hash(img, len(img), code, dg, C.byref(out), C.byref(olen)) == 0
    new = C.string_at(out, olen.value)
    C.CDLL(None).free(out)
    assert new[14] == code and len(new) == len(img) - 16 + (0 if code == 0 else LENGTHS[code])
    assert new[:14] == img[:14] and new[15:len(img) - 16] == img[15:len(img) - 16]
    assert B.decompress_buffer(new, host_threads=2) == data
    import lrz_decode
    assert bytes(lrz_decode.decode(new)) == data  # the independent Python decoder checks the digest with hashlib
    info = B.file_info(new)
    assert info.hash_code == code
    if code:
        bad = bytearray(new)
        bad[-1] ^= 1
        with pytest.raises(RuntimeError):
            B.decompress_buffer(bytes(bad))
        if code != 2 or lrz_decode.file_hash(2, b"") is not None:
            with pytest.raises(ValueError):
                lrz_decode.decode(bytes(bad))
        with pytest.raises(RuntimeError):
            B.decompress_buffer(new[:-1])  # a short digest
    # and back to MD5: the original image, byte for byte
    assert L.lrzgpu_set_file_hash(new, len(new), 1, hashlib.md5(data).digest(), C.byref(out), C.byref(olen)) == 0
    assert C.string_at(out, olen.value) == img
    C.CDLL(None).free(out)


def _magic(L, raw):
    m = Magic()
    rc = L.lrzgpu_read_magic(raw, len(raw), C.byref(m))
    return rc, m


def test_read_magic_current_version(B, O, L):
    data = datagen.text_like(200000, seed=7)
    img, _ = O.compress_buffer(data, compression_level=7, threads=2, processors=2, ramsize=RAM)
    rc, m = _magic(L, img)
    assert rc == 0 and (m.major, m.minor, m.magic_len) == (0, 14, 21)
    assert m.st_size == len(data) and m.enc_code == 0 and (m.hash_code, m.hash_len) == (1, 16)
    assert m.ctype == 1 and m.dict_size == 1 << 25 and img[18] == 26  # lzma2-style dictionary byte: (2 | (p & 1)) << (p / 2 + 11)
    assert bytes(m.lzma_properties)[0] == 0x5D and int.from_bytes(bytes(m.lzma_properties)[1:], "little") == m.dict_size
    assert (m.level, m.rzip_level) == (7, 7) and m.comment_length == 0 and m.filter_flag == 0


def _hdr(minor, n, **f):
    g = bytearray(n)
    g[0:4] = b"LRZI"
    g[5] = minor
    for k, v in f.items():
        at = int(k[1:])
        if isinstance(v, (bytes, bytearray)):
            g[at:at + len(v)] = v
        else:
            g[at] = v
    return bytes(g)


def test_read_magic_older_versions(L):
    """Headers laid out as doc/magic.header.txt describes the older formats and get_magic_v6 ... get_magic_v11 read them."""
    size = (123456789012).to_bytes(8, "little")
    props = bytes([0x5D, 0x00, 0x00, 0x00, 0x02])  # lc/lp/pb + a 32 MiB dictionary
    # 0.6: 24 bytes; lzma properties at 16-20, hash flag at 21, encryption at 22
    rc, m = _magic(L, _hdr(6, 24, b6=size, b16=props, b21=1))
    assert rc == 0 and (m.magic_len, m.st_size, m.hash_code, m.hash_len, m.enc_code) == (24, 123456789012, 1, 16, 0)
    assert m.ctype == 1 and m.dict_size == 1 << 25 and bytes(m.lzma_properties) == props
    rc, m = _magic(L, _hdr(6, 24, b6=bytes(range(1, 9)), b22=1))  # encrypted: the size field is the salt
    assert rc == 0 and m.enc_code == 1 and m.st_size == 0 and bytes(m.salt) == bytes(range(1, 9)) and m.costfactor == 1
    # 0.7: filter at 16 (delta: flag 7 in the low bits, offset - 1 in the high five), lzma at 17-21, hash 22, encryption 23
    rc, m = _magic(L, _hdr(7, 24, b6=size, b16=3, b17=props, b22=1))
    assert rc == 0 and (m.filter_flag, m.ctype, m.dict_size, m.hash_code) == (3, 1, 1 << 25, 1)
    rc, m = _magic(L, _hdr(7, 24, b6=size, b16=7 | (4 << 3)))
    assert rc == 0 and (m.filter_flag, m.delta) == (128, 5)
    rc, m = _magic(L, _hdr(7, 24, b6=size, b16=7 | (18 << 3)))
    assert (m.filter_flag, m.delta) == (128, 48)  # 17 -> 32, 18 -> 48, ...
    # 0.8: 18 bytes; hash 14, encryption 15, filter 16, byte 17: lzma2-style dictionary byte / zpaq / bzip3
    rc, m = _magic(L, _hdr(8, 18, b6=size, b14=5, b17=26))
    assert rc == 0 and (m.magic_len, m.hash_code, m.hash_len, m.ctype, m.dict_size) == (18, 5, 64, 1, 1 << 25)
    rc, m = _magic(L, _hdr(8, 18, b6=size, b17=0x80 | (5 << 4) | 7))
    assert (m.ctype, m.zpaq_level, m.zpaq_bs) == (2, 5, 7)
    rc, m = _magic(L, _hdr(8, 18, b6=size, b17=0xF0 | 6))
    assert (m.ctype, m.bzip3_bs) == (3, 6)
    # 0.9 / 0.10: two more bytes: levels and the comment length
    rc, m = _magic(L, _hdr(10, 20, b6=size, b14=1, b17=26, b18=(6 << 4) | 9, b19=5) + b"hello")
    assert rc == 0 and (m.magic_len, m.rzip_level, m.level, m.comment_length, m.comment) == (20, 6, 9, 5, b"hello")
    rc, m = _magic(L, _hdr(10, 20, b19=5) + b"hel")
    assert rc != 0  # the comment is cut short
    # 0.11+: compression type at 17, its property at 18, levels 19, comment 20; three encodings of delta by minor version
    rc, m = _magic(L, _hdr(11, 21, b6=size, b14=3, b16=7 | (0 << 3), b17=1, b18=26, b19=(7 << 4) | 7))
    assert rc == 0 and (m.hash_code, m.hash_len, m.filter_flag, m.delta, m.dict_size, m.level) == (3, 32, 128, 1, 1 << 25, 7)
    rc, m = _magic(L, _hdr(12, 21, b6=size, b16=7))
    assert (m.filter_flag, m.delta) == (7, 0)  # ARM64 in 0.12
    rc, m = _magic(L, _hdr(12, 21, b6=size, b16=17 << 3))
    assert (m.filter_flag, m.delta) == (128, 32)
    rc, m = _magic(L, _hdr(13, 21, b6=size, b16=8))
    assert (m.filter_flag, m.delta) == (8, 0)  # RISC-V in 0.13
    rc, m = _magic(L, _hdr(14, 21, b6=size, b16=128 + 20))
    assert (m.filter_flag, m.delta) == (128, 80)
    rc, m = _magic(L, _hdr(14, 21, b6=size, b17=(6 << 4) | 4, b18=15))
    assert rc == 0 and (m.ctype, m.zstd_strategy, m.zstd_level) == (4, 6, 15)
    rc, m = _magic(L, _hdr(14, 21, b6=size, b17=2, b18=(4 << 4) | 9))
    assert (m.ctype, m.zpaq_level, m.zpaq_bs) == (2, 4, 9)
    rc, m = _magic(L, _hdr(14, 21, b6=bytes([20, 1, 2, 3, 4, 5, 6, 7]), b15=2, b17=1, b18=26))
    assert rc == 0 and m.enc_code == 2 and m.costfactor == 20 and m.st_size == 0
    # refused: not an lrzip file, unknown versions, an invalid compression type, a header cut short
    assert _magic(L, b"LRZX" + bytes(30))[0] != 0
    assert _magic(L, _hdr(5, 24))[0] != 0 and _magic(L, _hdr(15, 21))[0] != 0
    assert _magic(L, _hdr(14, 21, b17=9))[0] != 0
    assert _magic(L, _hdr(14, 21)[:20])[0] != 0


def test_scan_access_hooks(B, O, L):
    """lrzgpu_full_tag / next_tag / match_len == single_full_tag / single_next_tag / single_match_len of src/rzip.c:
    the tag is the XOR of the frozen hash_index[] over 31 bytes, rolled one byte at a time; match lengths against a
    brute-force count; and the tags at the positions of the oracle's matches agree between source and destination."""
    hx = O.hash_index()
    L.lrzgpu_full_tag.restype = C.c_uint64
    L.lrzgpu_full_tag.argtypes = [C.c_char_p, C.c_int64]
    L.lrzgpu_next_tag.restype = C.c_uint64
    L.lrzgpu_next_tag.argtypes = [C.c_char_p, C.c_int64, C.c_uint64]
    L.lrzgpu_match_len.restype = C.c_int64
    L.lrzgpu_match_len.argtypes = [C.c_char_p, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.POINTER(C.c_int64)]
    data = datagen.long_range(200000, seed=12, base_frac=0.5, mutate_every=7001)
    n = len(data)

    def tag(p):
        t = 0
        for i in range(31):
            t ^= int(hx[data[p + i]])
        return t
    t = L.lrzgpu_full_tag(data, 0)
    assert t == tag(0)
    for p in range(1, 3000):
        t = L.lrzgpu_next_tag(data, p, t)
        if p % 97 == 0 or p < 40:
            assert t == tag(p) == L.lrzgpu_full_tag(data, p)
    rnd = random.Random(5)
    half = n // 2
    end = n - 31
    for _ in range(300):
        p0 = rnd.randrange(half + 100, end)
        op = p0 - half if rnd.random() < 0.7 else rnd.randrange(0, p0)
        last = rnd.choice([0, -1, p0 - 5, p0 - 40, half])
        f = 0
        while p0 + f < end and data[p0 + f] == data[op + f]:
            f += 1
        b = 0
        while p0 - b > max(0, last) and op - b > 0 and data[op - b - 1] == data[p0 - b - 1]:
            b += 1
        want = f + b if f + b >= 31 else 0
        rev = C.c_int64(-1)
        assert L.lrzgpu_match_len(data, p0, op, end, last, C.byref(rev)) == want and rev.value == b
    assert L.lrzgpu_match_len(data, 100, 100, end, 0, None) == 0 and L.lrzgpu_match_len(data, 100, 200, end, 0, None) == 0


def test_new_host_entry_points_survive_garbage(B, O, L):
    """Untrusted bytes through read_magic, set_file_hash and the decoder with every filter byte: an error code or a
    correct answer, never a crash."""
    rnd = random.Random(99)
    m = Magic()
    for _ in range(3000):
        n = rnd.choice([0, 5, 6, 17, 18, 20, 21, 24, 30, 300])
        raw = bytearray(rnd.getrandbits(8) for _ in range(n))
        if n >= 6 and rnd.random() < 0.8:
            raw[0:4] = b"LRZI"
            raw[4] = 0
            raw[5] = rnd.choice([5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15])
        L.lrzgpu_read_magic(bytes(raw), n, C.byref(m))
        out, olen = C.POINTER(C.c_ubyte)(), C.c_int64()
        if L.lrzgpu_set_file_hash(bytes(raw), n, rnd.randrange(-1, 16), bytes(64), C.byref(out), C.byref(olen)) == 0:
            C.CDLL(None).free(out)
    data = datagen.long_range(300000, seed=8)
    img, _ = O.compress_buffer(data, compression_level=7, threads=2, processors=2, ramsize=RAM)
    for filt in list(range(0, 12)) + [127, 128, 129, 144, 145, 159, 160, 255]:
        bad = bytearray(img)
        bad[16] = filt
        try:
            back = B.decompress_buffer(bytes(bad), host_threads=2)
            assert back == data  # (a filter that finds nothing to convert in these literals leaves them as they are)
        except RuntimeError:
            assert filt != 0  # changed literals: CRC / MD5 refuse; unknown filter bytes: refused outright
    for code in range(0, 20):
        bad = bytearray(img)
        bad[14] = code
        try:
            back = B.decompress_buffer(bytes(bad), host_threads=2)
            assert code == 1 and back == data
        except RuntimeError:
            assert code != 1
