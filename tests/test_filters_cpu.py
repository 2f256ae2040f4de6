"""BCJ / delta filters of the literal stream (src/stream.c:1587-1628 and 1926-1990): the library's
host implementations (lrzip-next_amd/csrc/filters.cpp) byte for byte against the reference's own converters
(src/lzma/C/Bra.c, Bra86.c, Delta.c compiled unmodified into oracle/_ref), in both directions, and the read side
undoing a filter named in magic[16]."""
import ctypes as C
import hashlib
import random
import struct

import pytest

import datagen

RAM = 80 * 100 * 1048576
X86, ARM, ARMT, PPC, SPARC, IA64, ARM64, RISCV, DELTA = 1, 2, 3, 4, 5, 6, 7, 8, 128
REF_NAME = {ARM: "ARM", ARMT: "ARMT", PPC: "PPC", SPARC: "SPARC", IA64: "IA64", ARM64: "ARM64", RISCV: "RISCV"}


@pytest.fixture(scope="module")
def R(O):
    r = O.ref_lzma()
    if r is None:
        pytest.skip("oracle/_ref/liblzma_ref.so missing")
    return r


def ref_filter(R, flag, delta, data, encode):
    """The reference's converter over one block, called as compthread / ucompthread call it (pc 0, fresh state)."""
    buf = C.create_string_buffer(bytes(data), len(data))
    if flag == DELTA:
        state = C.create_string_buffer(256)
        R.Delta_Init(state)
        f = R.Delta_Encode if encode else R.Delta_Decode
        f.argtypes = [C.c_char_p, C.c_uint, C.c_char_p, C.c_size_t]
        f.restype = None
        f(state, delta, buf, len(data))
    elif flag == X86:
        st = C.c_uint32(0)
        f = R.z7_BranchConvSt_X86_Enc if encode else R.z7_BranchConvSt_X86_Dec
        f.argtypes = [C.c_char_p, C.c_size_t, C.c_uint32, C.POINTER(C.c_uint32)]
        f.restype = C.c_void_p
        f(buf, len(data), 0, C.byref(st))
    else:
        f = getattr(R, "z7_BranchConv_%s_%s" % (REF_NAME[flag], "Enc" if encode else "Dec"))
        f.argtypes = [C.c_char_p, C.c_size_t, C.c_uint32]
        f.restype = C.c_void_p
        f(buf, len(data), 0)
    return buf.raw[:len(data)]


def lib_filter(B, flag, delta, data, encode):
    L = B.lib()
    L.lrzgpu_filter_block.argtypes = [C.c_int, C.c_int, C.c_char_p, C.c_int64, C.c_int]
    buf = C.create_string_buffer(bytes(data), len(data))
    assert L.lrzgpu_filter_block(flag, delta, buf, len(data), 1 if encode else 0) == 0
    return buf.raw[:len(data)]


def code_like(flag, n, seed):
    """Bytes with many instructions of the kind `flag` converts, in every shape its tests look at, among noise."""
    rnd = random.Random(seed)
    out = bytearray(rnd.getrandbits(8) for _ in range(n))
    if flag == X86:
        for _ in range(n // 6):
            i = rnd.randrange(max(1, n))
            out[i] = rnd.choice([0xE8, 0xE9, 0xE8, 0xE8, 0x00, 0xFF])
            if i + 4 < n and rnd.random() < 0.7:
                out[i + 4] = rnd.choice([0x00, 0xFF, 0x00, 0xFF, 0x01, 0xFE])
            if i + 3 < n and rnd.random() < 0.3:
                out[i + rnd.randrange(1, 4)] = rnd.choice([0xE8, 0xE9, 0x00, 0xFF])
    elif flag == ARM:
        for i in range(3, n, 4):
            if rnd.random() < 0.4:
                out[i] = 0xEB
    elif flag == ARMT:
        for i in range(0, n - 3, 2):
            if rnd.random() < 0.3:
                out[i + 1] = 0xF0 | rnd.randrange(8)
                out[i + 3] = 0xF8 | rnd.randrange(8)
            elif rnd.random() < 0.1:
                out[i + 1] = 0xF8 | rnd.randrange(8)  # a second half without a first
    elif flag == PPC:
        for i in range(0, n - 3, 4):
            if rnd.random() < 0.4:
                out[i] = 0x48 | rnd.randrange(4)
                out[i + 3] = (out[i + 3] & 0xFC) | rnd.choice([1, 1, 1, 0, 3])
    elif flag == SPARC:
        for i in range(0, n - 3, 4):
            r = rnd.random()
            if r < 0.25:
                out[i], out[i + 1] = 0x40, out[i + 1] & 0x3F
            elif r < 0.5:
                out[i], out[i + 1] = 0x7F, out[i + 1] | 0xC0
            elif r < 0.6:
                out[i] = rnd.choice([0x40, 0x7F])
    elif flag == ARM64:
        for i in range(0, n - 3, 4):
            r = rnd.random()
            if r < 0.3:
                struct.pack_into("<I", out, i, 0x94000000 | rnd.getrandbits(26))
            elif r < 0.7:  # ADRP with small, large, negative page offsets
                immhi = rnd.choice([rnd.getrandbits(19), rnd.getrandbits(15), 0x7FFFF ^ rnd.getrandbits(15), 0x78000, 0x7FFF, 0x8000, 0x77FFF])
                struct.pack_into("<I", out, i, 0x90000000 | rnd.getrandbits(2) << 29 | immhi << 5 | rnd.getrandbits(5))
    elif flag == IA64:
        for i in range(0, n - 15, 16):
            out[i] = (out[i] & 0xE0) | rnd.choice([0x10, 0x11, 0x12, 0x13, 0x16, 0x17, 0x18, 0x19, 0x1C, 0x1D, 0x00, 0x08])
            for slot in range(3):
                if rnd.random() < 0.5:  # make the slot a br.call: opcode 5 at bits 37..40, btype 0 at bits 9..11
                    bit = 5 + 41 * slot
                    v = int.from_bytes(out[i:i + 16], "little")
                    ins = (v >> bit) & ((1 << 41) - 1)
                    ins = (ins & ~(0xF << 37) & ~(0x7 << 9)) | (5 << 37)
                    v = (v & ~(((1 << 41) - 1) << bit)) | (ins << bit)
                    out[i:i + 16] = v.to_bytes(16, "little")
    elif flag == RISCV:
        i = 0
        while i + 8 <= n:
            r = rnd.random()
            if r < 0.25:  # JAL with assorted link registers (x1 and x5 are the calls)
                rd = rnd.choice([1, 5, 1, 5, 0, 2, 3, 9, 31])
                struct.pack_into("<I", out, i, (rnd.getrandbits(20) << 12) | (rd << 7) | 0x6F)
                i += 4
            elif r < 0.55:  # AUIPC + an instruction that may or may not read its rd; AUIPC x2 that looks like the marker
                rd = rnd.choice([1, 3, 5, 6, 7, 10, 17, 31, 0, 2, 2, 2])
                hi = rnd.getrandbits(20)
                if rd == 2 and rnd.random() < 0.6:
                    hi |= 3
                struct.pack_into("<I", out, i, (hi << 12) | (rd << 7) | 0x17)
                rs1 = rd if rnd.random() < 0.7 else rnd.randrange(32)
                op = rnd.choice([0x67, 0x13, 0x03, 0x23, 0x33, 0x01, 0x02])
                struct.pack_into("<I", out, i + 4, (rnd.getrandbits(12) << 20) | (rs1 << 15) | (rnd.getrandbits(3) << 12) | (rnd.getrandbits(5) << 7) | op)
                i += 8
            else:
                i += 2 if r < 0.7 else 4
    return bytes(out)


@pytest.mark.parametrize("flag", [X86, ARM, ARMT, PPC, SPARC, IA64, ARM64, RISCV])
def test_bcj_filters_equal_reference(B, R, flag):
    for seed, n in enumerate([0, 1, 2, 3, 4, 5, 6, 7, 8, 15, 16, 17, 31, 32, 33, 63, 64, 100, 1001, 4096, 65537, 300003]):
        for data in (code_like(flag, n, seed), datagen.KINDS["random"](n, seed=seed) if n else b"", bytes([0xE8, 0, 0, 0, 0] * (n // 5))[:n]):
            for enc in (True, False):
                assert lib_filter(B, flag, 0, data, enc) == ref_filter(R, flag, 0, data, enc), (flag, n, enc)
            assert lib_filter(B, flag, 0, lib_filter(B, flag, 0, data, True), False) == data


def test_x86_filter_on_dense_opcode_bytes(B, R):
    """The x86 converter's history of recent E8/E9 bytes: inputs made of nothing but opcode and sign bytes."""
    for seed in range(300):
        rnd = random.Random(seed)
        n = rnd.choice([5, 6, 7, 8, 9, 10, 11, 12, 13, 16, 20, 33, 100, 1000, 5000])
        alphabet = rnd.choice([[0xE8, 0xE9, 0, 0xFF, 1], [0xE8, 0, 0xFF], [0xE8, 0xE9, 0, 0xFF, 0x7F, 0x80, 0xFE, 2, 3, 4], [0xE8, 0xFF],
                               [0xE9, 0x00, 0x00, 0x00]])
        data = bytes(rnd.choice(alphabet) for _ in range(n))
        for enc in (True, False):
            assert lib_filter(B, X86, 0, data, enc) == ref_filter(R, X86, 0, data, enc), (seed, n, enc)


@pytest.mark.parametrize("delta", [1, 2, 3, 4, 7, 16, 32, 48, 255, 256])
def test_delta_filter_equals_reference(B, R, delta):
    for seed, n in enumerate([0, 1, 2, delta - 1, delta, delta + 1, 2 * delta, 1000, 65537]):
        data = datagen.KINDS["text"](n, seed=seed) if n else b""
        for enc in (True, False):
            assert lib_filter(B, DELTA, delta, data, enc) == ref_filter(R, DELTA, delta, data, enc), (delta, n, enc)
        assert lib_filter(B, DELTA, delta, lib_filter(B, DELTA, delta, data, True), False) == data


def test_unsupported_filters_are_refused(B):
    L = B.lib()
    L.lrzgpu_filter_block.argtypes = [C.c_int, C.c_int, C.c_char_p, C.c_int64, C.c_int]
    buf = C.create_string_buffer(64)
    for flag, delta in ((9, 0), (0, 0), (-1, 0), (127, 0), (DELTA, 0), (DELTA, 257)):
        assert L.lrzgpu_filter_supported(flag, delta) == 0
        assert L.lrzgpu_filter_block(flag, delta, buf, 64, 1) != 0


@pytest.mark.parametrize("flag,delta", [(X86, 0), (ARM, 0), (ARMT, 0), (PPC, 0), (SPARC, 0), (IA64, 0), (ARM64, 0), (RISCV, 0), (DELTA, 1), (DELTA, 4),
                                        (DELTA, 48)])
def test_read_side_undoes_the_filter(B, O, flag, delta):
    """A -n image (stored blocks) whose literal blocks were filtered one by one the way compthread does it and whose
    magic[16] names the filter decodes to the original; with the wrong flag it does not."""
    import lrz_decode
    L = B.lib()
    L.lrzgpu_set_file_filter.argtypes = [C.c_char_p, C.c_int64, C.c_int, C.c_int]
    data = code_like(flag if flag != DELTA else ARM, 3 * 1048576 + 5, seed=flag + delta)
    img, _ = O.compress_buffer(data, compression_level=7, threads=2, processors=2, ramsize=RAM, no_compress=True)
    hdr, chunks = lrz_decode.parse(img)
    new = bytearray(img)
    nblocks = 0
    for c in chunks:
        for blk in c["streams"][1]:  # (c_type, offset, c_len, u_len)
            c_type, off, c_len, u_len = blk[0], blk[1], blk[2], blk[3]
            assert c_type == 3 and c_len == u_len
            new[off:off + c_len] = lib_filter(B, flag, delta, bytes(new[off:off + c_len]), True)
            nblocks += 1
    assert nblocks >= 1
    buf = C.create_string_buffer(bytes(new), len(new))
    assert L.lrzgpu_set_file_filter(buf, len(new), flag, delta) == 0
    filtered = buf.raw[:len(new)]
    assert filtered[16] == (flag if flag != DELTA else 128 + (delta if delta <= 16 else delta // 16 + 15))
    assert B.decompress_buffer(filtered, host_threads=2) == data
    assert hashlib.md5(data).digest() == filtered[-16:]
    if bytes(new) != img:  # the filter changed something: without the flag the MD5 / CRC check must fail
        with pytest.raises(RuntimeError):
            B.decompress_buffer(bytes(new))
