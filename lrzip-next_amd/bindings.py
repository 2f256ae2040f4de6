"""ctypes bindings of the C ABI declared in include/lrzgpu.h (liblrzgpu.so).

This is plumbing for tests/ and bench.py: the product is the shared library.  There is no
CPU fallback here: if the library is missing, loading raises.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "liblrzgpu.so")


class Control(C.Structure):
    _fields_ = [("compression_level", C.c_int), ("rzip_compression_level", C.c_int), ("threads", C.c_int),
                ("processors", C.c_int), ("ramsize", C.c_int64), ("window", C.c_int64), ("dictSize", C.c_uint32),
                ("flags", C.c_uint32), ("threshold", C.c_int), ("device", C.c_int), ("host_threads", C.c_int),
                ("gpu_slots", C.c_int), ("verbose", C.c_int), ("st_size", C.c_int64),
                ("hash_resblock", C.c_uint8 * 16), ("lzma_properties", C.c_uint8 * 5), ("dictSize_used", C.c_uint32),
                ("stream_bufsize", C.c_int64), ("threads_used", C.c_int)]


class ScanStats(C.Structure):
    _fields_ = [(n, C.c_int64) for n in ("matches", "match_bytes", "literals", "literal_bytes", "inserts", "lookups",
                                          "tag_hits", "tag_misses", "hash_count", "tag_clean_ptr")] + \
               [("minimum_tag_mask", C.c_uint64), ("tag_mask", C.c_uint64)]


FLAG_NO_COMPRESS = 1 << 5
FLAG_THRESHOLD = 1 << 20
FLAG_NOBEMT = 1 << 27

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("liblrzgpu.so is not built (run __graft_entry__.build()); no fallback exists")
        L = C.CDLL(LIB_PATH)
        L.lrzgpu_version.restype = C.c_char_p
        L.lrzgpu_lz4_compress_default_size.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int]
        L.lrzgpu_lz4_compresses.argtypes = [C.c_char_p, C.c_int64, C.c_int, C.c_int]
        L.lrzgpu_lz4_compresses_dev.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_int]
        L.lrzgpu_lzma_match_lists.argtypes = [C.c_char_p, C.c_size_t, C.c_uint32, C.c_uint, C.c_uint, C.c_void_p,
                                              C.c_void_p, C.c_size_t, C.c_int]
        L.lrzgpu_lzma_match_lists.restype = C.c_int64
        L.lrzgpu_lzma_encode_with_lists.argtypes = [C.c_void_p, C.POINTER(C.c_size_t), C.c_char_p, C.c_size_t,
                                                    C.c_void_p, C.c_void_p, C.c_int, C.c_uint, C.c_int, C.c_int,
                                                    C.c_int, C.c_int]
        L.lrzgpu_LzmaCompress.argtypes = [C.c_void_p, C.POINTER(C.c_size_t), C.c_char_p, C.c_size_t, C.c_void_p,
                                          C.POINTER(C.c_size_t), C.c_int, C.c_uint, C.c_int, C.c_int, C.c_int,
                                          C.c_int, C.c_int]
        _lib = L
    return _lib


def lzma_match_lists(data: bytes, dict_size=1 << 25, fb=64, cut=48, device=0, per_pos=16):
    import numpy as np
    n = len(data)
    counts = np.zeros(max(n, 1), dtype=np.uint8)
    cap = n * per_pos + 4096
    pairs = np.zeros(cap, dtype=np.uint32)
    total = lib().lrzgpu_lzma_match_lists(data, n, dict_size, fb, cut, counts.ctypes.data, pairs.ctypes.data, cap, device)
    if total < 0:
        raise RuntimeError("lrzgpu_lzma_match_lists rc=%d" % total)
    return counts[:n], pairs[:total]


def lzma_encode_with_lists(data: bytes, counts, pairs, level=7, dict_size=1 << 25, fb=64, lc=3, lp=0, pb=2, cap=None):
    import numpy as np
    n = len(data)
    if cap is None:
        cap = n + n // 3 + 4096
    dst = C.create_string_buffer(cap)
    dlen = C.c_size_t(cap)
    counts = np.ascontiguousarray(counts, dtype=np.uint8)
    pairs = np.ascontiguousarray(pairs, dtype=np.uint32)
    if counts.size == 0:
        counts = np.zeros(1, dtype=np.uint8)
    if pairs.size == 0:
        pairs = np.zeros(1, dtype=np.uint32)
    rc = lib().lrzgpu_lzma_encode_with_lists(dst, C.byref(dlen), data, n, counts.ctypes.data, pairs.ctypes.data,
                                             level, dict_size, lc, lp, pb, fb)
    return rc, dst.raw[:dlen.value]


def lzma_compress(data: bytes, level=7, dict_size=1 << 25, fb=64, lc=3, lp=0, pb=2, threads=2, cap=None):
    n = len(data)
    if cap is None:
        cap = n + n // 3 + 4096
    dst = C.create_string_buffer(cap)
    dlen = C.c_size_t(cap)
    props = C.create_string_buffer(5)
    plen = C.c_size_t(5)
    rc = lib().lrzgpu_LzmaCompress(dst, C.byref(dlen), data, n, props, C.byref(plen), level, dict_size, lc, lp, pb,
                                   fb, threads)
    return rc, dst.raw[:dlen.value], props.raw
