"""ctypes bindings of the C ABI declared in include/lrzgpu.h (liblrzgpu.so).

This is plumbing for tests/ and bench.py: the product is the shared library.  There is no
CPU fallback here: if the library is missing, loading raises.
"""
import ctypes as C
import os

# Eight hardware queues for the scan, gate and finder streams (effective only if HIP is not yet initialised).  More is
# worse: with 32 the runtime holds more queues than the GPU keeps resident, the scheduler rotates them, and EVERY
# kernel of the process runs at about half its speed from the first step on (measured, round 6: k_resolve 392 ->
# 192 ms per launch, k_tag_scan alone 0.63 -> 1.10 TB/s, the step 21.9 -> 20.7 s); with 4 the streams queue up.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "liblrzgpu.so")


class Control(C.Structure):
    _fields_ = [("compression_level", C.c_int), ("rzip_compression_level", C.c_int), ("threads", C.c_int),
                ("processors", C.c_int), ("ramsize", C.c_int64), ("window", C.c_int64), ("dictSize", C.c_uint32),
                ("flags", C.c_uint32), ("threshold", C.c_int), ("device", C.c_int), ("host_threads", C.c_int),
                ("gpu_slots", C.c_int), ("verbose", C.c_int), ("st_size", C.c_int64),
                ("hash_resblock", C.c_uint8 * 16), ("lzma_properties", C.c_uint8 * 5), ("dictSize_used", C.c_uint32),
                ("stream_bufsize", C.c_int64), ("threads_used", C.c_int), ("zstd_level", C.c_int),
                ("scan_slots", C.c_int), ("eof", C.c_int),
                ("hash_code", C.c_int), ("filter_flag", C.c_int), ("delta", C.c_int), ("stdin_mode", C.c_int),
                ("stdout_mode", C.c_int), ("hash_full", C.c_uint8 * 64), ("fd_out", C.c_int),
                ("backoff_would_apply", C.c_int), ("malloc_probe", C.c_int)]


class ScanStats(C.Structure):
    _fields_ = [(n, C.c_int64) for n in ("matches", "match_bytes", "literals", "literal_bytes", "inserts", "lookups",
                                          "tag_hits", "tag_misses", "hash_count", "tag_clean_ptr")] + \
               [("minimum_tag_mask", C.c_uint64), ("tag_mask", C.c_uint64)]


FLAG_NO_COMPRESS = 1 << 5
FLAG_THRESHOLD = 1 << 20
FLAG_NOBEMT = 1 << 27
FLAG_ZSTD = 1 << 26

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
        L.lrzgpu_hash_search.argtypes = [C.c_char_p, C.c_int64, C.c_int, C.c_int, C.POINTER(C.c_int64),
                                         C.POINTER(C.POINTER(C.c_ubyte)), C.POINTER(C.c_int64), C.c_void_p,
                                         C.POINTER(C.c_int64), C.POINTER(C.c_uint32), C.POINTER(ScanStats), C.c_int]
        L.lrzgpu_hash_index.argtypes = [C.POINTER(C.c_uint64)]
        L.lrzgpu_control_init.argtypes = [C.POINTER(Control)]
        L.lrzgpu_compress_buffer.argtypes = [C.POINTER(Control), C.c_char_p, C.c_int64,
                                             C.POINTER(C.POINTER(C.c_ubyte)), C.POINTER(C.c_int64)]
        L.lrzgpu_compress_buffer_dev.argtypes = [C.POINTER(Control), C.c_void_p, C.c_int64,
                                                 C.POINTER(C.POINTER(C.c_ubyte)), C.POINTER(C.c_int64)]
        L.lrzgpu_compress_file.argtypes = [C.POINTER(Control), C.c_int, C.c_int]
        L.lrzgpu_rzip_fd.argtypes = [C.POINTER(Control), C.c_int, C.c_int]
        L.lrzgpu_plan.argtypes = [C.POINTER(Control), C.c_int64, C.POINTER(C.c_int64)]
        L.lrzgpu_container_store.argtypes = [C.POINTER(Control), C.c_int64, C.c_int, C.POINTER(C.c_int64),
                                             C.POINTER(C.c_char_p), C.POINTER(C.c_int64), C.POINTER(C.c_char_p),
                                             C.POINTER(C.c_int64), C.c_char_p, C.POINTER(C.POINTER(C.c_ubyte)),
                                             C.POINTER(C.c_int64)]
        L.lrzgpu_compress_chunks.argtypes = [C.POINTER(Control), C.c_char_p, C.c_int64, C.c_int, C.c_int,
                                             C.POINTER(C.c_int64), C.c_int, CHUNK_FN, C.c_void_p]
        L.lrzgpu_compress_chunks_dev.argtypes = [C.POINTER(Control), C.c_void_p, C.c_int64, C.c_int, C.c_int,
                                                 C.POINTER(C.c_int64), C.c_int, CHUNK_FN, C.c_void_p]
        L.lrzgpu_assemble_chunks.argtypes = [C.POINTER(Control), C.c_int64, C.c_int, C.POINTER(C.c_char_p),
                                             C.POINTER(C.c_int64), C.c_char_p, C.POINTER(C.POINTER(C.c_ubyte)),
                                             C.POINTER(C.c_int64)]
        L.lrzgpu_trim.restype = None
        _lib = L
    return _lib


class CompressThread(C.Structure):
    _fields_ = [("s_buf", C.c_void_p), ("c_type", C.c_uint8), ("s_len", C.c_int64), ("c_len", C.c_int64),
                ("sinfo", C.c_void_p), ("streamno", C.c_int)]


def lzma_compress_buf(block: bytes, ctl=None, **kw):
    """The back-end dispatch seam (lzma_compress_buf / zstd_compress_buf contract) on one block ->
    (return code, c_type, result bytes)."""
    c = ctl if ctl is not None else make_control(**kw)
    libc = C.CDLL(None)
    libc.malloc.restype = C.c_void_p
    libc.malloc.argtypes = [C.c_size_t]
    n = len(block)
    buf = libc.malloc(max(n, 1))
    C.memmove(buf, block, n)
    t = CompressThread(buf, 3, n, n, None, 1)
    f = lib().lrzgpu_lzma_compress_buf
    f.argtypes = [C.POINTER(Control), C.POINTER(CompressThread), C.c_int]
    rc = f(C.byref(c), C.byref(t), 0)
    out = C.string_at(t.s_buf, t.c_len) if rc == 0 else b""
    _LIBC_FREE(t.s_buf)
    return rc, t.c_type, out


def stream_out_file(path_out, chunks, st_size, **kw):
    """Drives the stream.h surface the way rzip_fd()/rzip_chunk()/hash_search() do: chunks = [(chunk_size,
    chunk_bytes, stream0 bytes, stream1 bytes)]; tokens are replayed through write_stream exactly in the order
    put_literal/put_match issue them (src/rzip.c:208-265).  Writes chunks at offset 21 of path_out (no magic, no
    hash) -> Control."""
    L = lib()
    c = make_control(**kw)
    c.st_size = st_size
    L.lrzgpu_open_stream_out.restype = C.c_void_p
    L.lrzgpu_open_stream_out.argtypes = [C.POINTER(Control), C.c_int, C.c_uint, C.c_int64, C.c_char]
    L.lrzgpu_write_stream.argtypes = [C.POINTER(Control), C.c_void_p, C.c_int, C.c_char_p, C.c_int64]
    L.lrzgpu_close_stream_out.argtypes = [C.POINTER(Control), C.c_void_p]
    fo = os.open(path_out, os.O_RDWR | os.O_CREAT | os.O_TRUNC, 0o644)
    try:
        os.write(fo, bytes(21))
        if L.lrzgpu_prepare_streamout_threads(C.byref(c)) != 0:
            raise RuntimeError("prepare_streamout_threads")
        for k, (chunk_size, cb, s0, s1) in enumerate(chunks):
            c.eof = 1 if k + 1 == len(chunks) else 0
            ss = L.lrzgpu_open_stream_out(C.byref(c), fo, 2, chunk_size, bytes([cb]))
            if not ss:
                raise RuntimeError("open_stream_out")
            i = lit = 0
            while i + 3 <= len(s0):
                head, ln = s0[i], s0[i + 1] | (s0[i + 2] << 8)
                if head == 0 and ln == 0:  # terminator: the CRC follows
                    rc = L.lrzgpu_write_stream(C.byref(c), ss, 0, s0[i:], len(s0) - i)
                    i = len(s0)
                elif head == 0:
                    rc = L.lrzgpu_write_stream(C.byref(c), ss, 0, s0[i:i + 3], 3)
                    rc = rc or L.lrzgpu_write_stream(C.byref(c), ss, 1, s1[lit:lit + ln], ln)
                    lit += ln
                    i += 3
                else:
                    rc = L.lrzgpu_write_stream(C.byref(c), ss, 0, s0[i:i + 3 + cb], 3 + cb)
                    i += 3 + cb
                if rc:
                    raise RuntimeError("write_stream rc=%d" % rc)
            if L.lrzgpu_close_stream_out(C.byref(c), ss) != 0:
                raise RuntimeError("close_stream_out")
        rc = L.lrzgpu_close_streamout_threads(C.byref(c))
        if rc != 0:
            raise RuntimeError("close_streamout_threads rc=%d" % rc)
        end = os.lseek(fo, 0, os.SEEK_CUR)
    finally:
        os.close(fo)
    return c, end


CHUNK_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_int64, C.c_int64, C.POINTER(C.c_ubyte), C.c_int64)


def compress_chunks(data=None, first=0, stride=1, victim_in=None, with_md5=False, dev_ptr=None, n=None, ctl=None, **kw):
    """The chunks k % stride == first of one file through the whole path ->
    ({chunk index: (victim_in, victim_out, chunk image bytes)}, Control).  data: host bytes, or dev_ptr + n."""
    c = ctl if ctl is not None else make_control(**kw)
    got = {}

    def on_chunk(_ctx, idx, vin, vout, img, ln):
        got[idx] = (vin, vout, C.string_at(img, ln) if ln < (1 << 31) - 1 else bytes(memoryview((C.c_ubyte * ln).from_address(C.addressof(img.contents))).cast("B")))
        return 0

    cb = CHUNK_FN(on_chunk)
    vi = None
    if victim_in is not None:
        vi = (C.c_int64 * len(victim_in))(*victim_in)
    if dev_ptr is not None:
        rc = lib().lrzgpu_compress_chunks_dev(C.byref(c), C.c_void_p(dev_ptr), n, first, stride, vi, int(with_md5), cb, None)
    else:
        rc = lib().lrzgpu_compress_chunks(C.byref(c), data, len(data), first, stride, vi, int(with_md5), cb, None)
    if rc != 0:
        raise RuntimeError("lrzgpu_compress_chunks rc=%d" % rc)
    return got, c


def assemble_chunks(images, st_size, md5, ctl=None, **kw):
    """magic + chunk images in order + the whole-input hash (`md5`: lrzgpu_hash_length(hash_code) bytes) -> .lrz bytes
    (host only)."""
    c = ctl if ctl is not None else make_control(**kw)
    need = lib().lrzgpu_hash_length(c.hash_code) if c.hash_code else 0
    assert need >= 0 and len(md5) >= need, "digest shorter than hash code %d needs (%d bytes)" % (c.hash_code, need)
    k = len(images)
    arr = (C.c_char_p * k)(*images)
    lens = (C.c_int64 * k)(*[len(i) for i in images])
    out = C.POINTER(C.c_ubyte)()
    olen = C.c_int64()
    rc = lib().lrzgpu_assemble_chunks(C.byref(c), st_size, k, arr, lens, bytes(md5), C.byref(out), C.byref(olen))
    if rc != 0:
        raise RuntimeError("lrzgpu_assemble_chunks rc=%d" % rc)
    return _take(out, olen), c


SHARD_COMPRESS_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int64), CHUNK_FN, C.c_void_p)


def compress_sharded_dev(dev_ptr, n, comm, ctl=None, chunk_ptrs=None, **kw):
    """lrzgpu_compress_sharded_dev (whole input on this rank's device) or, with chunk_ptrs (a list of device addresses,
    0 for chunks of other ranks), lrzgpu_compress_sharded_chunks_dev -> (LrzBuffer on rank 0 else None, Control, redone)."""
    c = ctl if ctl is not None else make_control(**kw)
    out = C.POINTER(C.c_ubyte)()
    olen = C.c_int64()
    redone = C.c_int64()
    if chunk_ptrs is not None:
        arr = (C.c_void_p * len(chunk_ptrs))(*[p or None for p in chunk_ptrs])
        rc = lib().lrzgpu_compress_sharded_chunks_dev(C.byref(c), arr, C.c_int64(n), C.byref(comm), C.byref(out), C.byref(olen), C.byref(redone))
    else:
        rc = lib().lrzgpu_compress_sharded_dev(C.byref(c), C.c_void_p(dev_ptr), C.c_int64(n), C.byref(comm), C.byref(out), C.byref(olen), C.byref(redone))
    if rc != 0:
        raise RuntimeError("lrzgpu_compress_sharded rc=%d" % rc)
    return (LrzBuffer(out, olen.value) if out else None), c, redone.value


def shard_protocol(n, comm, compress_fn, digest, ctl=None, **kw):
    """lrzgpu_shard_protocol over a Python per-rank compressor: compress_fn(first, stride, victim_in or None) ->
    {chunk index: (victim_in, victim_out, image bytes)}.  -> (.lrz bytes on rank 0 else None, redone)."""
    c = ctl if ctl is not None else make_control(**kw)
    n_seen = [0]

    def fn(_ctx, first, stride, victim_in, on_chunk, on_ctx):
        try:
            vi = None
            if victim_in:
                vi = [victim_in[k] for k in range(max(first + 1, stride))]
            for k, (vin, vout, img) in sorted(compress_fn(first, stride, vi).items()):
                buf = (C.c_ubyte * max(len(img), 1)).from_buffer_copy(img if img else b"\0")
                if on_chunk(on_ctx, k, vin, vout, buf, len(img)) != 0:
                    return -1
                n_seen[0] += 1
            return 0
        except Exception:
            import traceback
            traceback.print_exc()
            return -1

    cb = SHARD_COMPRESS_FN(fn)
    out = C.POINTER(C.c_ubyte)()
    olen = C.c_int64()
    redone = C.c_int64()
    rc = lib().lrzgpu_shard_protocol(C.byref(c), C.c_int64(n), C.byref(comm), cb, None, digest, C.byref(out), C.byref(olen), C.byref(redone))
    if rc != 0:
        raise RuntimeError("lrzgpu_shard_protocol rc=%d" % rc)
    return (_take(out, olen) if out else None), redone.value


def chunk_bytes_for(n):
    bits = 8
    while n >> bits > 0:
        bits += 1
    return bits // 8 + (1 if bits % 8 else 0)


def hash_search(data: bytes, level=7, chunk_bytes=None, victim_round=0, device=0):
    """One rzip chunk through the GPU scan -> (stream0, stream1, stats, crc, victim_round_out)."""
    n = len(data)
    if chunk_bytes is None:
        chunk_bytes = chunk_bytes_for(n)
    s0 = C.POINTER(C.c_ubyte)()
    s0len = C.c_int64()
    s1 = C.create_string_buffer(max(n, 1))
    s1len = C.c_int64()
    crc = C.c_uint32()
    st = ScanStats()
    vr = C.c_int64(victim_round)
    rc = lib().lrzgpu_hash_search(data, n, level, chunk_bytes, C.byref(vr), C.byref(s0), C.byref(s0len), s1,
                                  C.byref(s1len), C.byref(crc), C.byref(st), device)
    if rc != 0:
        raise RuntimeError("lrzgpu_hash_search rc=%d" % rc)
    stream0 = C.string_at(s0, s0len.value)
    C.CDLL(None).free(s0)
    return stream0, s1.raw[:s1len.value], st, crc.value, vr.value


def tag_candidates_dev(ptr: int, n: int, first=0, min_mask=1, reps=1, only_tags=False, device=0):
    """K1 of the scan alone over a chunk in HBM -> (count, checksum, ms per pass): lrzgpu_tag_candidates_dev."""
    cnt = C.c_int64()
    chk = C.c_uint64()
    ms = C.c_double()
    f = lib().lrzgpu_tag_candidates_dev
    f.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_uint64, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_uint64),
                  C.POINTER(C.c_double), C.c_int, C.c_int]
    rc = f(C.c_void_p(ptr), n, first, min_mask, reps, C.byref(cnt), C.byref(chk), C.byref(ms), 1 if only_tags else 0, device)
    if rc != 0:
        raise RuntimeError("lrzgpu_tag_candidates_dev rc=%d" % rc)
    return cnt.value, chk.value, ms.value


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


def lzma_match_lists_prefix(data: bytes, block_n, dict_size=1 << 25, fb=64, cut=48, device=0, per_pos=16):
    """lrzgpu_lzma_match_lists_prefix: the BT4 finder on a prefix of a block of block_n bytes (hash mask of the block)."""
    import numpy as np
    n = len(data)
    counts = np.zeros(max(n, 1), dtype=np.uint8)
    cap = n * per_pos + 4096
    pairs = np.zeros(cap, dtype=np.uint32)
    f = lib().lrzgpu_lzma_match_lists_prefix
    f.argtypes = [C.c_char_p, C.c_size_t, C.c_size_t, C.c_uint32, C.c_uint, C.c_uint, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    f.restype = C.c_int64
    total = f(data, n, block_n, dict_size, fb, cut, counts.ctypes.data, pairs.ctypes.data, cap, device)
    if total < 0:
        raise RuntimeError("lrzgpu_lzma_match_lists_prefix rc=%d" % total)
    return counts[:n], pairs[:total]


def lzma_match_lists_hc5(data: bytes, dict_size=1 << 22, fb=32, cut=16, device=0, per_pos=40):
    import numpy as np
    n = len(data)
    counts = np.zeros(max(n, 1), dtype=np.uint8)
    cap = n * per_pos + 4096
    pairs = np.zeros(cap, dtype=np.uint32)
    f = lib().lrzgpu_lzma_match_lists_hc5
    f.argtypes = lib().lrzgpu_lzma_match_lists.argtypes
    f.restype = C.c_int64
    total = f(data, n, dict_size, fb, cut, counts.ctypes.data, pairs.ctypes.data, cap, device)
    if total < 0:
        raise RuntimeError("lrzgpu_lzma_match_lists_hc5 rc=%d" % total)
    return counts[:n], pairs[:total]


def tail_flags(data: bytes, counts, pairs):
    """What lzma_mf.hip k_gather computes per pair: do the two bytes after the match and one literal continue at
    the same distance?  (numpy; for feeding the flagged / packed list formats without a GPU)"""
    import numpy as np
    a = np.frombuffer(data, dtype=np.uint8)
    n = len(a)
    counts = np.asarray(counts, dtype=np.int64)
    pos = np.repeat(np.arange(n, dtype=np.int64), counts[:n] // 2)
    ln = pairs[0::2].astype(np.int64)
    d1 = pairs[1::2].astype(np.int64) + 1
    at = pos + ln + 1
    ok = at + 2 <= n
    at = np.where(ok, at, 0)
    src = np.where(ok, at - d1, 0)
    return ok & (a[at] == a[src]) & (a[np.minimum(at + 1, n - 1)] == a[np.minimum(src + 1, n - 1)])


def format_lists(data: bytes, counts, pairs, list_format):
    """plain (len, dist-1) couples -> list format 1 (flag in bit 31 of len) or 2 (one packed word per pair)."""
    import numpy as np
    if list_format == 0:
        return pairs
    f = tail_flags(data, counts, pairs).astype(np.uint32)
    if list_format == 1:
        out = np.array(pairs, dtype=np.uint32, copy=True)
        out[0::2] |= f << np.uint32(31)
        return out
    return (f << np.uint32(31)) | ((pairs[0::2] - np.uint32(2)) << np.uint32(25)) | pairs[1::2]


def lzma_encode_with_lists(data: bytes, counts, pairs, level=7, dict_size=1 << 25, fb=64, lc=3, lp=0, pb=2, cap=None, list_format=0):
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
    if list_format:
        f = lib().lrzgpu_lzma_encode_with_lists_fmt
        f.argtypes = [C.c_void_p, C.POINTER(C.c_size_t), C.c_char_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                      C.c_uint, C.c_int, C.c_int, C.c_int, C.c_int]
        rc = f(dst, C.byref(dlen), data, n, counts.ctypes.data, pairs.ctypes.data, list_format, level, dict_size, lc, lp, pb, fb)
    else:
        rc = lib().lrzgpu_lzma_encode_with_lists(dst, C.byref(dlen), data, n, counts.ctypes.data, pairs.ctypes.data,
                                                 level, dict_size, lc, lp, pb, fb)
    return rc, dst.raw[:dlen.value]


def lzma_encode_with_lists_staged(data: bytes, counts, pairs, early_positions, level=7, dict_size=1 << 25, fb=64, lc=3, lp=0, pb=2,
                                  cap=None, list_format=0, early_counts=None, early_pairs=None, stage_step=0):
    """lrzgpu_lzma_encode_with_lists_staged: the parser started on the lists of the first early_positions positions,
    handed stage_step more every time it asks (0 = the rest in one piece)."""
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
    f = lib().lrzgpu_lzma_encode_with_lists_staged
    f.argtypes = [C.c_void_p, C.POINTER(C.c_size_t), C.c_char_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int,
                  C.c_uint, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t]
    f.restype = C.c_int
    ec = ep = None
    if early_counts is not None:
        ec = np.ascontiguousarray(early_counts, dtype=np.uint8)
        ep = np.ascontiguousarray(early_pairs, dtype=np.uint32)
        if ep.size == 0:
            ep = np.zeros(1, dtype=np.uint32)
        assert ec.size >= min(early_positions, n)
    rc = f(dst, C.byref(dlen), data, n, counts.ctypes.data, pairs.ctypes.data, early_positions, list_format, level, dict_size, lc, lp, pb, fb,
           ec.ctypes.data if ec is not None else None, ep.ctypes.data if ep is not None else None, stage_step)
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


def make_control(level=7, rzip_level=0, threads=1, processors=1, ramsize=80 * 100 * 1048576, window=0, dict_size=0,
                 no_compress=False, lz4_test=True, threshold=100, nobemt=False, device=0, host_threads=0,
                 gpu_slots=0, verbose=0, zstd=False, zstd_level=0, scan_slots=0, hash_code=None, filter_flag=None,
                 delta=0, stdin_mode=False, stdout_mode=False, malloc_probe=False):
    c = Control()
    lib().lrzgpu_control_init(C.byref(c))
    c.compression_level = level
    c.rzip_compression_level = rzip_level
    c.threads = threads
    c.processors = processors
    c.ramsize = ramsize
    c.window = window
    c.dictSize = dict_size
    c.flags = (FLAG_NO_COMPRESS if no_compress else 0) | (FLAG_THRESHOLD if lz4_test else 0) | \
              (FLAG_NOBEMT if nobemt else 0) | (FLAG_ZSTD if zstd else 0)
    c.threshold = threshold
    c.device = device
    c.host_threads = host_threads
    c.gpu_slots = gpu_slots
    c.verbose = verbose
    c.zstd_level = zstd_level
    c.scan_slots = scan_slots
    if hash_code is not None:
        c.hash_code = hash_code
    if filter_flag is not None:
        c.filter_flag = filter_flag
        c.delta = delta
    c.stdin_mode = 1 if stdin_mode else 0
    c.stdout_mode = 1 if stdout_mode else 0
    c.malloc_probe = 1 if malloc_probe else 0
    return c


def _take(out, olen):
    n = olen.value
    if n < (1 << 31) - 1:
        res = C.string_at(out, n)
    else:  # string_at() takes a C int
        res = bytes(memoryview((C.c_ubyte * n).from_address(C.addressof(out.contents))).cast("B"))
    C.CDLL(None).free(out)
    return res


_LIBC_FREE = C.CDLL(None).free
_LIBC_FREE.argtypes = [C.c_void_p]


class LrzBuffer:
    """The malloc()ed .lrz image a compress call returned, without a copy into Python bytes."""

    def __init__(self, ptr, size):
        self._ptr, self.size = ptr, size

    def __len__(self):
        return self.size

    def view(self):
        return memoryview((C.c_ubyte * self.size).from_address(C.addressof(self._ptr.contents))).cast("B")

    def tobytes(self):
        return C.string_at(self._ptr, self.size)

    def free(self):
        if self._ptr is not None:
            _LIBC_FREE(self._ptr)
            self._ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:  # interpreter shutdown: the process is going away anyway
            pass


def compress_buffer(data: bytes, **kw):
    """Whole-file compress of a host buffer -> (.lrz bytes, Control)."""
    c = make_control(**kw)
    out = C.POINTER(C.c_ubyte)()
    olen = C.c_int64()
    rc = lib().lrzgpu_compress_buffer(C.byref(c), data, len(data), C.byref(out), C.byref(olen))
    if rc != 0:
        raise RuntimeError("lrzgpu_compress_buffer rc=%d" % rc)
    return _take(out, olen), c


def compress_device(ptr: int, n: int, ctl=None, copy=True, **kw):
    """Whole-file compress of a buffer resident in HBM (ptr = device address) -> (.lrz bytes, Control);
    copy=False returns the library's own buffer as an LrzBuffer instead of bytes."""
    c = ctl if ctl is not None else make_control(**kw)
    out = C.POINTER(C.c_ubyte)()
    olen = C.c_int64()
    rc = lib().lrzgpu_compress_buffer_dev(C.byref(c), C.c_void_p(ptr), n, C.byref(out), C.byref(olen))
    if rc != 0:
        raise RuntimeError("lrzgpu_compress_buffer_dev rc=%d" % rc)
    if not copy:
        return LrzBuffer(out, olen.value), c
    return _take(out, olen), c


def decompress_buffer(img, host_threads=0):
    """The library's own round-trip verifier: .lrz image (bytes or LrzBuffer) -> original bytes."""
    if isinstance(img, LrzBuffer):
        ptr, n = C.cast(img._ptr, C.c_void_p), img.size
    else:
        ptr, n = C.cast(C.c_char_p(img), C.c_void_p), len(img)
    out = C.POINTER(C.c_ubyte)()
    olen = C.c_int64()
    f = lib().lrzgpu_decompress_buffer
    f.argtypes = [C.c_void_p, C.c_int64, C.POINTER(C.POINTER(C.c_ubyte)), C.POINTER(C.c_int64), C.c_int]
    rc = f(ptr, n, C.byref(out), C.byref(olen), host_threads)
    if rc != 0:
        raise RuntimeError("lrzgpu_decompress_buffer rc=%d" % rc)
    return _take(out, olen)


class Info(C.Structure):
    _fields_ = [("major", C.c_int), ("minor", C.c_int), ("st_size", C.c_int64), ("compressed_size", C.c_int64),
                ("hash_code", C.c_int), ("lzma", C.c_int), ("dict_prop", C.c_int), ("level", C.c_int),
                ("rzip_level", C.c_int), ("chunks", C.c_int64), ("blocks", C.c_int64), ("blocks_lzma", C.c_int64),
                ("stream_c_len", C.c_int64 * 2), ("stream_u_len", C.c_int64 * 2)]


class Magic(C.Structure):
    _fields_ = [("major", C.c_int), ("minor", C.c_int), ("magic_len", C.c_int), ("st_size", C.c_int64), ("enc_code", C.c_int),
                ("salt", C.c_uint8 * 8), ("costfactor", C.c_int), ("hash_code", C.c_int), ("hash_len", C.c_int),
                ("filter_flag", C.c_int), ("delta", C.c_int), ("ctype", C.c_int), ("dict_size", C.c_uint32),
                ("lzma_properties", C.c_uint8 * 5), ("zpaq_bs", C.c_int), ("zpaq_level", C.c_int), ("bzip3_bs", C.c_int),
                ("zstd_strategy", C.c_int), ("zstd_level", C.c_int), ("level", C.c_int), ("rzip_level", C.c_int),
                ("comment_length", C.c_int), ("comment", C.c_char * 256)]


def read_magic(img: bytes):
    m = Magic()
    f = lib().lrzgpu_read_magic
    f.argtypes = [C.c_char_p, C.c_int64, C.POINTER(Magic)]
    rc = f(img, len(img), C.byref(m))
    if rc != 0:
        raise RuntimeError("lrzgpu_read_magic rc=%d" % rc)
    return m


def file_info(img: bytes):
    info = Info()
    f = lib().lrzgpu_file_info
    f.argtypes = [C.c_char_p, C.c_int64, C.POINTER(Info)]
    rc = f(img, len(img), C.byref(info))
    if rc != 0:
        raise RuntimeError("lrzgpu_file_info rc=%d" % rc)
    return info


def compress_file(path_in, path_out, rzip_only_fd=False, **kw):
    """lrzgpu_compress_file (magic + chunks + MD5) or, with rzip_only_fd, lrzgpu_rzip_fd (chunks + MD5 at
    the current offset of fd_out, the caller writes the magic) -> Control."""
    c = make_control(**kw)
    fi = os.open(path_in, os.O_RDONLY)
    fo = os.open(path_out, os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644)
    try:
        rc = (lib().lrzgpu_rzip_fd if rzip_only_fd else lib().lrzgpu_compress_file)(C.byref(c), fi, fo)
    finally:
        os.close(fi)
        os.close(fo)
    if rc != 0:
        raise RuntimeError("compress via fd rc=%d" % rc)
    return c


def decompress_file(path_in, path_out, host_threads=0):
    fi = os.open(path_in, os.O_RDONLY)
    fo = os.open(path_out, os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644)
    try:
        rc = lib().lrzgpu_decompress_file(fi, fo, host_threads)
    finally:
        os.close(fi)
        os.close(fo)
    if rc != 0:
        raise RuntimeError("lrzgpu_decompress_file rc=%d" % rc)


def plan(st_size, **kw):
    c = make_control(**kw)
    chunk = C.c_int64()
    rc = lib().lrzgpu_plan(C.byref(c), st_size, C.byref(chunk))
    if rc != 0:
        raise RuntimeError("lrzgpu_plan rc=%d" % rc)
    return c, chunk.value


def container_store(st_size, chunk_sizes, streams0, streams1, md5, **kw):
    c = make_control(**kw)
    n = len(chunk_sizes)
    cs = (C.c_int64 * n)(*chunk_sizes)
    s0 = (C.c_char_p * n)(*streams0)
    s0l = (C.c_int64 * n)(*[len(x) for x in streams0])
    s1 = (C.c_char_p * n)(*streams1)
    s1l = (C.c_int64 * n)(*[len(x) for x in streams1])
    out = C.POINTER(C.c_ubyte)()
    olen = C.c_int64()
    rc = lib().lrzgpu_container_store(C.byref(c), st_size, n, cs, s0, s0l, s1, s1l, md5, C.byref(out), C.byref(olen))
    if rc != 0:
        raise RuntimeError("lrzgpu_container_store rc=%d" % rc)
    return _take(out, olen)
