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
