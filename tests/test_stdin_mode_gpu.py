"""STDIN / STDOUT modes (src/rzip.c:800-836, 970-973, 1014-1017, 1041-1107, src/util.c:179-188,
src/stream.c:1725-1729, src/lrzip.c:141-144): chunks of max_mmap bytes cut the way mmap_stdin() cuts them -- an input
that ends exactly on a chunk boundary is followed by an empty last chunk --, blocks sized from the first chunk, a sixth
of the RAM and a size-less magic when writing to stdout.  Product image == oracle image, byte for byte; decodes."""
import os
import threading

import pytest

import datagen

pytestmark = pytest.mark.gpu
PAGE = 4096


def modes(B, O, data, ram, si, so, **kw):
    want, fs = O.compress_buffer(data, compression_level=1, threads=2, processors=2, ramsize=ram, workers=4,
                                 stdin_mode=si, stdout_mode=so, **{k: int(v) for k, v in kw.items()})
    got, ctl = B.compress_buffer(data, level=1, threads=2, processors=2, ramsize=ram, host_threads=4,
                                 stdin_mode=si, stdout_mode=so, **kw)
    assert got == want, (len(data), ram, si, so, kw)
    assert B.decompress_buffer(got, host_threads=2) == data
    return got, fs


@pytest.mark.parametrize("si,so", [(1, 0), (0, 1), (1, 1)])
def test_stdin_stdout_images_equal_oracle(B, O, si, so):
    ram = 60 << 20                      # maxram 20 MiB (10 MiB to stdout): the STDIN chunk size; -L1 fits that
    chunk = (ram // (6 if so else 3)) // PAGE * PAGE
    base = datagen.long_range(chunk, seed=5)
    for n in (0, 1, 4095, 100000, chunk - 1, chunk, chunk + 1, 2 * chunk, 2 * chunk + 12345):
        data = (base * 3)[:n]
        got, fs = modes(B, O, data, ram, si, so)
        if si:
            # full chunks, then the short (possibly empty) one that noticed the end
            assert fs.n_chunks == n // chunk + 1
        size_in_magic = int.from_bytes(got[6:14], "little")
        assert size_in_magic == (0 if (so and fs.n_chunks > 1) else n)
    # and stored blocks (-n): maxram is not halved for the second buffer
    modes(B, O, (base * 2)[:chunk + 777], ram, si, so, no_compress=True)


def test_stdin_mode_differs_from_file_mode_where_the_reference_does(B, O):
    """Same bytes, same flags: as a file the input is one chunk sized from st_size; from STDIN several."""
    ram = 60 << 20
    data = datagen.text_like(50 << 20, seed=9)
    as_file, _ = B.compress_buffer(data, level=1, threads=2, processors=2, ramsize=ram, host_threads=4)
    want_file, fs_file = O.compress_buffer(data, compression_level=1, threads=2, processors=2, ramsize=ram, workers=4)
    as_stdin, fs_stdin = modes(B, O, data, ram, 1, 0)
    assert as_file == want_file
    assert fs_stdin.n_chunks == 3 and fs_file.n_chunks == 2 and as_file != as_stdin


def test_pipe_into_compress_file_with_stdin_mode(B, O, tmp_path):
    import ctypes as C
    ram = 60 << 20
    data = datagen.long_range(25 << 20, seed=11)
    want, _ = O.compress_buffer(data, compression_level=1, threads=2, processors=2, ramsize=ram, workers=4, stdin_mode=1)
    r, w = os.pipe()

    def feed():
        with os.fdopen(w, "wb") as f:
            f.write(data)

    t = threading.Thread(target=feed)
    t.start()
    out = tmp_path / "piped.lrz"
    fo = os.open(out, os.O_RDWR | os.O_CREAT | os.O_TRUNC, 0o644)
    c = B.make_control(level=1, threads=2, processors=2, ramsize=ram, host_threads=4, stdin_mode=True)
    try:
        assert B.lib().lrzgpu_compress_file(C.byref(c), r, fo) == 0
    finally:
        os.close(r)
        os.close(fo)
        t.join()
    assert out.read_bytes() == want
    assert c.st_size == len(data)
