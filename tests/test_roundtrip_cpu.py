"""The independent decoder (tests/lrz_decode.py) against the oracle's .lrz images: pins the decoder
that the GPU round-trip tests rely on, and the oracle's container from the reading side."""
import pytest

import datagen
import lrz_decode

RAM = 80 * 100 * 1048576


@pytest.mark.parametrize("kind", ["text", "longrange", "random", "zeros", "phrases", "sparse"])
def test_decode_oracle_image(O, kind):
    data = datagen.KINDS[kind](2 * 1048576 + 333, seed=9)
    img, fs = O.compress_buffer(data, compression_level=7, threads=4, processors=4, ramsize=RAM)
    assert bytes(lrz_decode.decode(img)) == data


def test_decode_stored_and_empty(O):
    data = datagen.text_like(300000, seed=3)
    img, _ = O.compress_buffer(data, compression_level=7, threads=2, processors=2, ramsize=RAM, no_compress=True)
    hdr, chunks = lrz_decode.parse(img)
    assert not hdr["lzma"] and all(b[0] == 3 for c in chunks for s in c["streams"] for b in s)
    assert bytes(lrz_decode.decode(img)) == data
    img, _ = O.compress_buffer(b"", compression_level=7, threads=2, processors=2, ramsize=RAM)
    assert len(lrz_decode.decode(img)) == 0


def test_decoder_rejects_corruption(O):
    data = datagen.long_range(1048576, seed=4)
    img, _ = O.compress_buffer(data, compression_level=7, threads=2, processors=2, ramsize=RAM)
    bad = bytearray(img)
    bad[len(bad) // 2] ^= 0x40
    with pytest.raises(Exception):
        lrz_decode.decode(bytes(bad))


@pytest.mark.parametrize("level", [1, 3, 5, 7, 9])
@pytest.mark.parametrize("kind", ["text", "longrange", "random", "zeros", "phrases", "few"])
def test_library_decoder_on_oracle_images(B, O, kind, level):
    """lrzgpu_decompress_buffer (host LZMA decoder + token replay + CRC/MD5 checks) against images
    written by the oracle (reference LzmaCompress inside): the library's read side is independent of
    its write side, and agrees with the Python decoder."""
    data = datagen.KINDS[kind](1048576 + 4321, seed=level + 20)
    img, _ = O.compress_buffer(data, compression_level=level, threads=4, processors=4, ramsize=RAM)
    assert B.decompress_buffer(img, host_threads=4) == data
    assert bytes(lrz_decode.decode(img)) == data


def test_library_decoder_edge_cases(B, O):
    for data, kw in ((b"", {}), (b"x", {}), (datagen.text_like(300000, seed=3), {"no_compress": True}),
                     (datagen.long_range(31, seed=1), {}), (bytes(70000), {})):
        img, _ = O.compress_buffer(data, compression_level=7, threads=2, processors=2, ramsize=RAM, **kw)
        assert B.decompress_buffer(img) == data
    img, _ = O.compress_buffer(datagen.long_range(1 << 20, seed=4), compression_level=7, threads=2, processors=2, ramsize=RAM)
    for cut in (20, 100, len(img) // 2, len(img) - 1):
        with pytest.raises(RuntimeError):
            B.decompress_buffer(img[:cut])
    for flip in (200, len(img) // 3, len(img) // 2, len(img) - 5):  # payload bytes and the MD5 trailer
        bad = bytearray(img)
        bad[flip] ^= 0x21
        with pytest.raises(RuntimeError):
            B.decompress_buffer(bytes(bad))
    # stored blocks: nothing but the chunk CRC and the MD5 notices a changed literal -- both are checked by the thread
    # that follows the rebuild (images of 1 MiB and more), and a clean image still decodes after a bad one
    data = datagen.text_like(3 * 524288 + 5, seed=8)
    img, _ = O.compress_buffer(data, compression_level=7, threads=2, processors=2, ramsize=RAM, no_compress=True)
    for flip in (len(img) // 2, len(img) - 3):
        bad = bytearray(img)
        bad[flip] ^= 0x01
        with pytest.raises(RuntimeError):
            B.decompress_buffer(bytes(bad))
    assert B.decompress_buffer(img) == data
    assert B.decompress_buffer(img, host_threads=1) == data  # (checks inline)


def test_file_info_and_decompress_file(B, O, tmp_path):
    data = datagen.long_range(3 * 1048576 + 17, seed=6)
    img, fs = O.compress_buffer(data, compression_level=7, threads=4, processors=4, ramsize=RAM)
    info = B.file_info(img)
    hdr, chunks = lrz_decode.parse(img)
    assert (info.major, info.minor, info.st_size, info.compressed_size) == (0, 14, len(data), len(img))
    assert info.hash_code == 1 and info.lzma == 1 and info.level == 7 and info.chunks == len(chunks) == fs.n_chunks
    assert info.blocks == sum(len(s) for c in chunks for s in c["streams"]) == fs.n_blocks
    assert info.stream_u_len[1] == sum(b[3] for c in chunks for b in c["streams"][1])
    src, dst = tmp_path / "a.lrz", tmp_path / "a.out"
    src.write_bytes(img)
    B.decompress_file(str(src), str(dst))
    assert dst.read_bytes() == data


@pytest.mark.parametrize("zl", [0, 3, 15, 19])
def test_zstd_images_decode(B, O, zl):
    """--zstd back end (system libzstd): oracle image -> both decoders; magic carries strategy and level."""
    data = datagen.long_range(2 * 1048576 + 99, seed=8) + datagen.text_like(1 << 20, seed=9)
    img, fs = O.compress_buffer(data, compression_level=6, threads=4, processors=4, ramsize=RAM, zstd=1, zstd_level=zl)
    hdr, chunks = lrz_decode.parse(img)
    assert hdr["zstd"] and not hdr["lzma"] and hdr["zstd_level"] == (zl or 15)
    assert any(b[0] == 10 for c in chunks for s in c["streams"] for b in s)
    assert bytes(lrz_decode.decode(img)) == data
    assert B.decompress_buffer(img, host_threads=4) == data


def test_library_decoder_survives_fuzz(B, O):
    """400 random mutations (flips, truncations, inserted junk) of small images: every call returns
    an error or the exact original -- never a crash, never different bytes."""
    import random
    data = datagen.long_range(200000, seed=12) + datagen.text_like(100000, seed=13)
    imgs = [O.compress_buffer(data, compression_level=lvl, threads=2, processors=2, ramsize=RAM)[0] for lvl in (3, 7)]
    imgs.append(O.compress_buffer(data, compression_level=5, threads=2, processors=2, ramsize=RAM, zstd=1)[0])
    r = random.Random(2024)
    for k in range(400):
        img = bytearray(r.choice(imgs))
        kind = r.randrange(4)
        if kind == 0:
            for _ in range(r.randrange(1, 4)):
                img[r.randrange(len(img))] ^= 1 << r.randrange(8)
        elif kind == 1:
            img = img[:r.randrange(len(img))]
        elif kind == 2:
            pos = r.randrange(len(img))
            img[pos:pos] = bytes(r.randrange(256) for _ in range(r.randrange(1, 9)))
        else:
            pos = r.randrange(21, len(img))
            img[pos:pos + 8] = (r.getrandbits(64)).to_bytes(8, "little")
        try:
            out = B.decompress_buffer(bytes(img), host_threads=2)
        except RuntimeError:
            continue
        assert out == data, k


def _crafted_image(cb, c_type, c_len, u_len, nxt, st_size=64, payload=b"\0" * 64):
    """A minimal .lrz whose first stream-0 block header carries the given (possibly absurd) lengths."""
    le = lambda v: (v & ((1 << (8 * cb)) - 1)).to_bytes(cb, "little")
    magic = bytearray(21)
    magic[0:4] = b"LRZI"
    magic[4], magic[5] = 0, 14
    magic[6:14] = st_size.to_bytes(8, "little")
    magic[14] = 1
    magic[17] = 1
    img = bytes(magic) + bytes([cb, 1]) + le(st_size)
    img += bytes([3]) + le(0) + le(0) + le(2 * (1 + 3 * cb))  # stream 0 initial header -> first block
    img += bytes([3]) + le(0) + le(0) + le(0)                  # stream 1 initial header, empty
    img += bytes([c_type]) + le(c_len) + le(u_len) + le(nxt) + payload + b"\0" * 16
    return img


@pytest.mark.parametrize("c_type", [3, 6, 10])
def test_library_decoder_rejects_wrapping_lengths(B, c_type):
    """Lengths near 2^64 in an 8-byte-wide header must be rejected, not wrap past the bounds checks
    (the image is untrusted input): error return, no crash, no allocation of the wrapped size."""
    big = (1 << 64) - 64
    cases = [
        _crafted_image(8, c_type, big, big, 0),            # c_len wraps b.off + c_len
        _crafted_image(8, c_type, 64, big, 0),             # u_len wraps the running total
        _crafted_image(8, c_type, 64, 64, big),            # next-header offset wraps base + nxt
        _crafted_image(8, c_type, 64, (1 << 63) + 5, 0),   # u_len far beyond st_size
        _crafted_image(8, 3, 16, 64, 0),                   # stored block with c_len != u_len
    ]
    for img in cases:
        with pytest.raises(RuntimeError):
            B.decompress_buffer(img, host_threads=2)
        try:  # the -i walk sees the same headers: error or figures, never a crash
            B.file_info(img)
        except RuntimeError:
            pass
