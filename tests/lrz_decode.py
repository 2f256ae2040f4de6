"""Independent .lrz (lrzip-next 0.14, LZMA / stored blocks, MD5 trailer) decoder: test infrastructure.

Written from the container description in SURVEY.md / DESIGN.md (reference: src/stream.c
open_stream_in / fill_buffer, src/runzip.c unzip_literal / unzip_match), with Python's liblzma
binding as the LZMA decoder.  It gives the parity tests a size-independent property:
decode(compress(x)) == x, MD5 trailer == md5(x), per-chunk CRC == crc32(chunk)."""
import hashlib
import lzma
import zlib
from concurrent.futures import ThreadPoolExecutor

import numpy as np


def _val(buf, pos, n):
    return int.from_bytes(bytes(buf[pos:pos + n]), "little")


def _dict_size(prop):
    return 0xFFFFFFFF if prop == 40 else (2 | (prop & 1)) << (prop // 2 + 11)


def _lzma_raw(payload, u_len, dict_size):
    d = lzma.LZMADecompressor(format=lzma.FORMAT_RAW,
                              filters=[{"id": lzma.FILTER_LZMA1, "dict_size": dict_size, "lc": 3, "lp": 0, "pb": 2}])
    out = d.decompress(bytes(payload), max_length=u_len)
    if len(out) != u_len:
        raise ValueError("LZMA block decoded to %d bytes, header says %d" % (len(out), u_len))
    return out


_ZSTD = None


def _zstd_raw(payload, u_len):
    """zstd blocks of --zstd files: the system libzstd through ctypes."""
    global _ZSTD
    import ctypes as C
    if _ZSTD is None:
        _ZSTD = C.CDLL("libzstd.so.1")
        _ZSTD.ZSTD_decompress.restype = C.c_size_t
        _ZSTD.ZSTD_decompress.argtypes = [C.c_void_p, C.c_size_t, C.c_char_p, C.c_size_t]
        _ZSTD.ZSTD_isError.argtypes = [C.c_size_t]
    out = C.create_string_buffer(u_len if u_len else 1)
    src = bytes(payload)
    r = _ZSTD.ZSTD_decompress(out, u_len, src, len(src))
    if _ZSTD.ZSTD_isError(r) or r != u_len:
        raise ValueError("zstd block failed to decode to %d bytes" % u_len)
    return out.raw[:u_len]


HASH_LEN = [4, 16, 20, 32, 48, 64, 32, 64, 16, 32, 64, 16, 32, 64]


def file_hash(code, data):
    """hashlib's value of hash `code` (None for RIPEMD-160 where this Python's OpenSSL does not offer it)."""
    if code == 2:
        try:
            return hashlib.new("ripemd160", data).digest()
        except (ValueError, TypeError):
            return None
    if code >= 8:
        return (hashlib.shake_128 if code <= 10 else hashlib.shake_256)(data).digest(HASH_LEN[code])
    return {1: hashlib.md5, 3: hashlib.sha256, 4: hashlib.sha384, 5: hashlib.sha512, 6: hashlib.sha3_256, 7: hashlib.sha3_512}[code](data).digest()


def parse(lrz):
    """-> (header dict, [chunk dict]) with every block located but not yet decoded."""
    buf = memoryview(lrz)
    if bytes(buf[:4]) != b"LRZI" or buf[4] != 0 or buf[5] != 14:
        raise ValueError("not an lrzip-next 0.14 file")
    hdr = {"st_size": _val(buf, 6, 8), "md5": buf[14] == 1, "lzma": buf[17] == 1, "zstd": (buf[17] & 15) == 4,
           "zstd_strategy": buf[17] >> 4, "zstd_level": buf[18] if (buf[17] & 15) == 4 else 0,
           "dict_size": _dict_size(buf[18]) if buf[17] == 1 else 0, "levels": buf[19], "comment": buf[20]}
    pos = 21 + hdr["comment"]
    chunks = []
    while True:
        cb, eof = buf[pos], buf[pos + 1]
        size = _val(buf, pos + 2, cb)
        base = pos + 2 + cb
        hlen = 1 + 3 * cb
        end = base + 2 * hlen
        streams = []
        for s in range(2):
            blocks = []
            h = base + s * hlen
            while True:
                c_type, c_len, u_len, nxt = buf[h], _val(buf, h + 1, cb), _val(buf, h + 1 + cb, cb), _val(buf, h + 1 + 2 * cb, cb)
                if c_len:
                    blocks.append((c_type, h + hlen, c_len, u_len))
                    end = max(end, h + hlen + c_len)
                else:
                    end = max(end, h + hlen)
                if not nxt:
                    break
                h = base + nxt
            streams.append(blocks)
        chunks.append({"chunk_bytes": cb, "eof": eof, "size": size, "streams": streams})
        pos = end
        if eof:
            break
    # the hash after the last chunk: code in magic[14] (`hashes[]`, src/main.c:64-79); MD5 is the default
    hdr["hash_code"] = buf[14]
    hlen = HASH_LEN[buf[14]] if 1 <= buf[14] < len(HASH_LEN) else 0
    hdr["hash_digest"] = bytes(buf[pos:pos + hlen]) if hlen else None
    hdr["md5_digest"] = hdr["hash_digest"] if hdr["md5"] else None
    hdr["end"] = pos + hlen
    hdr["filter"] = buf[16]
    return hdr, chunks


def _stream_bytes(buf, blocks, dict_size, pool):
    def one(b):
        c_type, off, c_len, u_len = b
        if c_type == 3:
            if c_len != u_len:
                raise ValueError("stored block with c_len != u_len")
            return bytes(buf[off:off + c_len])
        if c_type == 6:
            return _lzma_raw(buf[off:off + c_len], u_len, dict_size)
        if c_type == 10:
            return _zstd_raw(buf[off:off + c_len], u_len)
        raise ValueError("block type %d is outside this decoder" % c_type)
    return b"".join(pool.map(one, blocks))


def _unrzip(s0, s1, cb, out):
    """Replay the token stream (src/runzip.c:146-260) into out (np.uint8[chunk]); returns the stored CRC."""
    s1 = np.frombuffer(s1, dtype=np.uint8)
    i = lit = cur = 0
    n0 = len(s0)
    while True:
        t = s0[i]
        ln = s0[i + 1] | (s0[i + 2] << 8)
        i += 3
        if t == 0:
            if ln == 0:
                break
            out[cur:cur + ln] = s1[lit:lit + ln]
            lit += ln
            cur += ln
        else:
            ofs = int.from_bytes(s0[i:i + cb], "little")
            i += cb
            src = cur - ofs
            if src < 0 or ofs == 0:
                raise ValueError("match reaches before the chunk")
            done = 0
            while done < ln:  # overlapping copies replicate the period
                k = min(ln - done, cur + done - src)
                out[cur + done:cur + done + k] = out[src:src + k]
                done += k
            cur += ln
    if i + 4 != n0 or lit != len(s1):
        raise ValueError("trailing bytes in the rzip streams")
    return cur, int.from_bytes(s0[i:i + 4], "big")


def decode(lrz, threads=8):
    """-> original bytes (np.uint8 array); raises on any inconsistency (CRC, MD5, sizes)."""
    buf = memoryview(lrz)
    hdr, chunks = parse(lrz)
    out = np.empty(hdr["st_size"], dtype=np.uint8)
    at = 0
    with ThreadPoolExecutor(threads) as pool:
        for c in chunks:
            s0 = _stream_bytes(buf, c["streams"][0], hdr["dict_size"], pool)
            s1 = _stream_bytes(buf, c["streams"][1], hdr["dict_size"], pool)
            n, crc = _unrzip(s0, s1, c["chunk_bytes"], out[at:])
            if (zlib.crc32(out[at:at + n]) & 0xFFFFFFFF) != crc:
                raise ValueError("chunk CRC mismatch")
            at += n
    if at != hdr["st_size"]:
        raise ValueError("decoded %d bytes, header says %d" % (at, hdr["st_size"]))
    if hdr["filter"]:
        raise ValueError("filtered literal blocks: this decoder has no converters (the library's read side does)")
    if hdr["hash_digest"] is not None:
        want = file_hash(hdr["hash_code"], out)
        if want is not None and want != hdr["hash_digest"]:
            raise ValueError("hash trailer mismatch")
    if hdr["end"] != len(lrz):
        raise ValueError("bytes after the MD5 trailer")
    return out
