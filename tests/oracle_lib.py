"""ctypes bindings for the ORACLE (oracle/liblrzo.so, oracle/_ref/liblzma_ref.so).

Test infrastructure only: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  Never imported by the product package.
"""
import ctypes as C
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ODIR = os.path.join(ROOT, "oracle")


class Level(C.Structure):
    _fields_ = [("mb_used", C.c_ulong), ("initial_freq", C.c_uint), ("max_chain_len", C.c_uint)]


class RzipStats(C.Structure):
    _fields_ = [(n, C.c_int64) for n in ("matches", "match_bytes", "literals", "literal_bytes",
                                          "tag_hits", "tag_misses", "inserts", "lookups", "hash_count")] + \
               [("minimum_tag_mask", C.c_uint64), ("tag_mask", C.c_uint64), ("tag_clean_ptr", C.c_int64)]


class Params(C.Structure):
    _fields_ = [("compression_level", C.c_int), ("rzip_level", C.c_int), ("no_compress", C.c_int),
                ("threads", C.c_int), ("processors", C.c_int), ("ramsize", C.c_int64), ("window", C.c_int64),
                ("lz4_test", C.c_int), ("threshold", C.c_int), ("nobemt", C.c_int), ("dict_size", C.c_uint32),
                ("workers", C.c_int), ("verbose", C.c_int), ("zstd", C.c_int), ("zstd_level", C.c_int),
                ("file_size", C.c_int64), ("stdin_mode", C.c_int), ("stdout_mode", C.c_int), ("filter_flag", C.c_int),
                ("malloc_probe", C.c_int)]


class FileStats(C.Structure):
    _fields_ = [("stream_bufsize", C.c_int64), ("threads_used", C.c_int), ("dict_size", C.c_uint32),
                ("n_chunks", C.c_int64), ("n_blocks", C.c_int64), ("blocks_lzma", C.c_int64),
                ("blocks_none", C.c_int64), ("rz", RzipStats)]


PUT0 = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(C.c_ubyte), C.c_int64)
PUT1 = C.CFUNCTYPE(None, C.c_void_p, C.c_int64, C.c_int64)


class Sink(C.Structure):
    _fields_ = [("ctx", C.c_void_p), ("put0", PUT0), ("put1", PUT1)]


def build():
    """(Re)build the oracle libraries; safe to call repeatedly."""
    subprocess.run(["make", "-s", "-C", ODIR], check=True, stdout=subprocess.DEVNULL)


_lib = None
_ref = None


def lib():
    global _lib
    if _lib is None:
        p = os.path.join(ODIR, "liblrzo.so")
        if not os.path.exists(p):
            build()
        L = C.CDLL(p)
        L.lrzo_hash_index.argtypes = [C.POINTER(C.c_uint64)]
        L.lrzo_rzip_chunk_table.argtypes = [C.c_char_p, C.c_int64, C.c_int, C.c_int, C.POINTER(C.c_uint64),
                                            C.POINTER(C.c_int64), C.POINTER(Sink), C.POINTER(RzipStats),
                                            C.POINTER(C.c_uint32), C.c_void_p]
        L.lrzo_crc32.argtypes = [C.c_uint32, C.c_char_p, C.c_size_t]
        L.lrzo_crc32.restype = C.c_uint32
        L.lrzo_lz4_compress_default_size.argtypes = [C.c_char_p, C.c_int, C.c_int]
        L.lrzo_lz4_compresses.argtypes = [C.c_char_p, C.c_int64, C.c_int]
        L.lrzo_lzma_mf_bt4.argtypes = [C.c_char_p, C.c_size_t, C.c_uint32, C.c_uint, C.c_uint,
                                       C.c_void_p, C.c_void_p, C.c_size_t]
        L.lrzo_lzma_mf_bt4.restype = C.c_int64
        L.lrzo_lzma_mf_hc5.argtypes = [C.c_char_p, C.c_size_t, C.c_uint32, C.c_uint, C.c_uint,
                                       C.c_void_p, C.c_void_p, C.c_size_t]
        L.lrzo_lzma_mf_hc5.restype = C.c_int64
        L.lrzo_lzma_hash_mask.argtypes = [C.c_uint32, C.c_uint64]
        L.lrzo_lzma_hash_mask.restype = C.c_uint32
        L.lrzo_params_default.argtypes = [C.POINTER(Params)]
        L.lrzo_compress_buffer.argtypes = [C.POINTER(Params), C.c_void_p, C.c_int64, C.c_void_p,
                                           C.POINTER(C.POINTER(C.c_ubyte)), C.POINTER(C.c_int64),
                                           C.POINTER(FileStats)]
        L.lrzo_plan.argtypes = [C.POINTER(Params), C.c_int64, C.POINTER(FileStats)]
        L.lrzo_rzip_level.argtypes = [C.c_int]
        L.lrzo_rzip_level.restype = C.POINTER(Level)
        _lib = L
    return _lib


def ref_lzma():
    """oracle/_ref/liblzma_ref.so : the reference's own LZMA sources, compiled unmodified."""
    global _ref
    if _ref is None:
        p = os.path.join(ODIR, "_ref", "liblzma_ref.so")
        if not os.path.exists(p):
            build()
        if not os.path.exists(p):
            return None
        R = C.CDLL(p)
        R.LzmaCompress.argtypes = [C.c_void_p, C.POINTER(C.c_size_t), C.c_char_p, C.c_size_t, C.c_void_p,
                                   C.POINTER(C.c_size_t), C.c_int, C.c_uint, C.c_int, C.c_int, C.c_int,
                                   C.c_int, C.c_int]
        R.LzmaUncompress.argtypes = [C.c_void_p, C.POINTER(C.c_size_t), C.c_char_p, C.POINTER(C.c_size_t),
                                     C.c_char_p, C.c_size_t]
        _ref = R
    return _ref


def hash_index():
    a = (C.c_uint64 * 256)()
    lib().lrzo_hash_index(a)
    return list(a)


def rzip_chunk(data: bytes, level=7, chunk_bytes=None, victim_round=0, want_table=False):
    """Returns (stream0 bytes, stream1 bytes, stats, crc, victim_round_out[, table])."""
    import numpy as np
    n = len(data)
    if chunk_bytes is None:
        bits = 8
        while n >> bits > 0:
            bits += 1
        chunk_bytes = bits // 8 + (1 if bits % 8 else 0)
    s0 = bytearray()
    runs = []

    def put0(ctx, p, ln):
        s0.extend(C.string_at(p, ln))

    def put1(ctx, off, ln):
        runs.append((off, ln))

    sink = Sink(None, PUT0(put0), PUT1(put1))
    hx = (C.c_uint64 * 256)()
    lib().lrzo_hash_index(hx)
    vr = C.c_int64(victim_round)
    st = RzipStats()
    crc = C.c_uint32()
    table = None
    tptr = None
    if want_table:
        lv = lib().lrzo_rzip_level(level).contents
        hashsize = lv.mb_used * (1048576 // 16)
        bits = 0
        while (1 << bits) < hashsize:
            bits += 1
        table = np.zeros(2 << bits, dtype=np.uint64)
        tptr = table.ctypes.data
    lib().lrzo_rzip_chunk_table(data, n, level, chunk_bytes, hx, C.byref(vr), C.byref(sink), C.byref(st),
                                C.byref(crc), tptr)
    s1 = b"".join(data[o:o + ln] for o, ln in runs)
    out = (bytes(s0), s1, st, crc.value, vr.value)
    return out + (table,) if want_table else out


def lzma_compress_ref(data: bytes, level=7, dict_size=1 << 25, fb=None, threads=2, lc=3, lp=0, pb=2):
    R = ref_lzma()
    if fb is None:
        fb = 32 if level < 7 else 64
    cap = len(data) + len(data) // 3 + 4096
    dst = C.create_string_buffer(cap)
    dlen = C.c_size_t(cap)
    props = C.create_string_buffer(5)
    plen = C.c_size_t(5)
    rc = R.LzmaCompress(dst, C.byref(dlen), data, len(data), props, C.byref(plen), level, dict_size,
                        lc, lp, pb, fb, threads)
    return rc, dst.raw[:dlen.value], props.raw


def lzma_uncompress_ref(comp: bytes, props: bytes, out_len: int):
    R = ref_lzma()
    dst = C.create_string_buffer(max(out_len, 1))
    dlen = C.c_size_t(out_len)
    slen = C.c_size_t(len(comp))
    rc = R.LzmaUncompress(dst, C.byref(dlen), comp, C.byref(slen), props, 5)
    return rc, dst.raw[:dlen.value]


FILTER_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_size_t)
_FILTER_NAMES = {2: "ARM", 3: "ARMT", 4: "PPC", 5: "SPARC", 6: "IA64", 7: "ARM64", 8: "RISCV"}


def ref_filter(flag, delta=0, encode=True):
    """The reference's own converter for one block (oracle/_ref: Bra.c, Bra86.c, BraIA64.c, Delta.c compiled
    unmodified), called the way compthread / ucompthread call it (src/stream.c:1587-1628, 1926-1990): pc 0,
    fresh x86 / delta state.  Returns a python callable (address, length)."""
    R = ref_lzma()
    sfx = "Enc" if encode else "Dec"
    if flag == 1:
        f = getattr(R, "z7_BranchConvSt_X86_" + sfx)
        f.argtypes = [C.c_void_p, C.c_size_t, C.c_uint32, C.POINTER(C.c_uint32)]
        f.restype = C.c_void_p

        def run(p, n):
            st = C.c_uint32(0)  # Z7_BRANCH_CONV_ST_X86_STATE_INIT_VAL
            f(p, n, 0, C.byref(st))
    elif flag == 128:
        init, conv = R.Delta_Init, (R.Delta_Encode if encode else R.Delta_Decode)
        init.argtypes = [C.c_void_p]
        conv.argtypes = [C.c_void_p, C.c_uint, C.c_void_p, C.c_size_t]

        def run(p, n):
            st = C.create_string_buffer(256)  # DELTA_STATE_SIZE
            init(st)
            conv(st, delta, p, n)
    else:
        f = getattr(R, "z7_BranchConv_%s_%s" % (_FILTER_NAMES[flag], sfx))
        f.argtypes = [C.c_void_p, C.c_size_t, C.c_uint32]
        f.restype = C.c_void_p

        def run(p, n):
            f(p, n, 0)
    return run


def magic16(flag, delta=0):
    """write_magic()'s filter byte, src/lrzip.c:146-156"""
    if flag == 128:
        return 128 + (delta if delta <= 16 else (delta >> 4) + 15)
    return flag


_keep_filter = []


_memo = []  # the last few answers for large byte strings: several GPU tests ask for the same oracle image (tens of CPU-seconds each)


def compress_buffer(data, filter_flag=0, filter_delta=0, **kw):
    """Whole-file oracle compress -> (.lrz bytes, FileStats).  data: bytes, or a numpy uint8 array
    (no copy: multi-GiB inputs).  filter_flag / filter_delta: run the reference's converter over every
    literal block first (oracle/_ref) and say so in magic[16]."""
    key = None
    if isinstance(data, (bytes, bytearray)) and len(data) >= (32 << 20) and not filter_flag:
        import hashlib
        key = (hashlib.blake2b(data, digest_size=16).digest(), len(data), tuple(sorted((k, v) for k, v in kw.items() if k != "workers")))
        for k, res, fs in _memo:
            if k == key:
                return res, fs
    res, fs = _compress_buffer(data, filter_flag, filter_delta, **kw)
    if key is not None and len(res) <= (256 << 20):
        _memo.append((key, res, fs))
        del _memo[:-3]
    return res, fs


def _compress_buffer(data, filter_flag=0, filter_delta=0, **kw):
    L = lib()
    p = Params()
    L.lrzo_params_default(C.byref(p))
    for k, v in kw.items():
        if not hasattr(p, k):
            raise KeyError(k)
        setattr(p, k, v)
    L.lrzo_set_filter.argtypes = [C.c_void_p]
    if filter_flag:
        run = ref_filter(filter_flag, filter_delta, True)
        cb = FILTER_FN(lambda ptr, n: run(ptr, n))
        _keep_filter[:] = [cb]
        L.lrzo_set_filter(C.cast(cb, C.c_void_p))
        p.filter_flag = magic16(filter_flag, filter_delta)
    else:
        L.lrzo_set_filter(None)
    fn = None
    if p.zstd and not p.no_compress:
        z = C.CDLL("libzstd.so.1")  # the system library, as the reference links it
        L.lrzo_set_zstd.argtypes = [C.c_void_p]
        L.lrzo_set_zstd(C.cast(z.ZSTD_compress, C.c_void_p))
    elif not p.no_compress:
        R = ref_lzma()
        fn = C.cast(R.LzmaCompress, C.c_void_p)
    out = C.POINTER(C.c_ubyte)()
    olen = C.c_int64()
    fs = FileStats()
    if isinstance(data, (bytes, bytearray)):
        keep = bytes(data)  # alive until the call returns
        src, n = C.cast(C.c_char_p(keep), C.c_void_p), len(keep)
    else:
        src, n = C.c_void_p(data.ctypes.data), int(data.size)
    rc = L.lrzo_compress_buffer(C.byref(p), src, n, fn, C.byref(out), C.byref(olen), C.byref(fs))
    if rc != 0:
        raise RuntimeError("lrzo_compress_buffer rc=%d" % rc)
    if olen.value < (1 << 31) - 1:
        res = C.string_at(out, olen.value)
    else:  # string_at() takes a C int
        res = bytes(memoryview((C.c_ubyte * olen.value).from_address(C.addressof(out.contents))).cast("B"))
    C.CDLL(None).free(out)
    return res, fs


def mf_hc5(data: bytes, dict_size=1 << 22, fb=32, cut=16):
    """Per-position match lists of the HC5 finder (levels 1-4) -> (offsets u64[n+1], pairs u32[])."""
    import numpy as np
    n = len(data)
    offs = np.zeros(n + 1, dtype=np.uint64)
    total = lib().lrzo_lzma_mf_hc5(data, n, dict_size, fb, cut, offs.ctypes.data, None, 0)
    if total < 0:
        raise RuntimeError("hc5 oracle failed %d" % total)
    pairs = np.zeros(max(total, 1), dtype=np.uint32)
    lib().lrzo_lzma_mf_hc5(data, n, dict_size, fb, cut, offs.ctypes.data, pairs.ctypes.data, int(total))
    return offs, pairs[:total]


def mf_bt4(data: bytes, dict_size=1 << 25, fb=64, cut=48):
    """Per-position final match lists (MT BT4 semantics) -> (offsets u64[n+1], pairs u32[])."""
    import numpy as np
    n = len(data)
    offs = np.zeros(n + 1, dtype=np.uint64)
    total = lib().lrzo_lzma_mf_bt4(data, n, dict_size, fb, cut, offs.ctypes.data, None, 0)
    if total < 0:
        raise RuntimeError("mf oracle failed %d" % total)
    pairs = np.zeros(max(total, 1), dtype=np.uint32)
    lib().lrzo_lzma_mf_bt4(data, n, dict_size, fb, cut, offs.ctypes.data, pairs.ctypes.data, int(total))
    return offs, pairs[:total]
