"""BASELINE.json's configurations on the GPU, each at the flags SURVEY 8(d) maps it to.  The default sizes keep
the whole GPU suite inside minutes and are compared byte for byte with the oracle; the full-size cfg 5 (32 GiB of
noise) and a 2.5 GiB cfg 4 run as property tests -- block types, sizes, CRC/MD5 and a round trip through the
library's decoder (tools/full_configs.sh: the 10 GiB cfg 4 and the full 32 GiB decode, log under profiles/)."""
import ctypes as C
import hashlib
import os

import pytest

import datagen

pytestmark = pytest.mark.gpu

FULL = os.environ.get("LRZGPU_FULL_CONFIGS") == "1"
RAM = 80 * 100 << 20


def _note(line):
    """timings of the full-size runs for profiles/ (LRZGPU_TIMES_FILE)"""
    f = os.environ.get("LRZGPU_TIMES_FILE")
    if f:
        with open(f, "a") as o:
            o.write(line + "\n")


def test_cfg1_text_rzip_only_whole_file(B, O):
    """cfg 1: 256 MiB of text, -n (rzip only, blocks stored), -w 3 -p1: one chunk, the whole-file path."""
    data = datagen.text_alnum(256 << 20, seed=11)
    want, fs = O.compress_buffer(data, no_compress=1, threads=1, processors=1, ramsize=RAM, window=3)
    assert fs.n_chunks == 1
    got, ctl = B.compress_buffer(data, no_compress=True, threads=1, processors=1, ramsize=RAM, window=3)
    assert bytes(ctl.hash_resblock) == hashlib.md5(data).digest()
    assert got == want
    info = B.file_info(got)
    assert info.chunks == 1 and info.blocks_lzma == 0 and info.st_size == len(data)


def test_cfg4_source_tree_tar_zstd15(B, O):
    """cfg 4 at reduced size: tar of copies of one source tree, --zstd --zstd-level 15 (=> rzip level 6), several
    chunks; zstd blocks through the host libzstd like the reference."""
    data = datagen.source_tree_tar(12, 20 << 20, seed=7)  # ~ 250 MiB, tree distance 20 MiB
    assert len(data) > 2 * 104857600
    kw = dict(level=7, threads=8, processors=16, ramsize=RAM, window=1, zstd=True, zstd_level=15)
    want, fs = O.compress_buffer(data, compression_level=7, threads=8, processors=16, ramsize=RAM, window=1, zstd=1, zstd_level=15, workers=16)
    assert fs.n_chunks >= 3
    got, ctl = B.compress_buffer(data, host_threads=16, **kw)
    assert got == want
    assert got[17] == (6 << 4) + 4 and got[18] == 15 and got[19] >> 4 == 6  # strategy 6 | zstd, level 15, rzip level 6
    assert B.decompress_buffer(got, host_threads=16) == data


def test_cfg5_random_lz4_early_out(B, O):
    """cfg 5 at 512 MiB (32 GiB below): incompressible input, every literal block must come out stored (CTYPE 3) through the lz4
    gate; -L7, one chunk."""
    import numpy as np
    data = np.random.default_rng(5).integers(0, 256, size=1 << 29, dtype=np.uint8).tobytes()
    kw = dict(level=7, threads=16, processors=16, ramsize=24 << 30)
    want, fs = O.compress_buffer(data, compression_level=7, threads=16, processors=16, ramsize=24 << 30, workers=16)
    got, ctl = B.compress_buffer(data, host_threads=16, **kw)
    assert got == want
    info = B.file_info(got)
    assert info.chunks == 1 and info.blocks_lzma <= 1 and info.stream_c_len[1] == info.stream_u_len[1]


def test_cfg5_full_32gib_random(B):
    import torch
    n = 32 << 30
    # seeded 64-bit PRNG on the GPU: splitmix64 of a counter (three multiply-xorshift rounds, elementwise), 1 GiB a time
    import time
    t_gen = time.time()
    buf = torch.empty(n + 256, dtype=torch.uint8, device="cuda")
    step = 1 << 30
    m30, m27, m31 = (1 << 34) - 1, (1 << 37) - 1, (1 << 33) - 1
    c1, c2, inc = -4658895280553007687, -7723592293110705685, -7046029254386353131  # 0xBF58476D1CE4E5B9, 0x94D049BB133111EB, 0x9E3779B97F4A7C15
    for o in range(0, n, step):
        x = (torch.arange(o // 8 + 5, o // 8 + 5 + step // 8, device="cuda", dtype=torch.int64)) * inc
        x = (x ^ ((x >> 30) & m30)) * c1
        x = (x ^ ((x >> 27) & m27)) * c2
        x = x ^ ((x >> 31) & m31)
        buf[o:o + step] = x.view(torch.uint8)
        del x
    buf[n:] = 0
    torch.cuda.synchronize()
    ram = os.sysconf("SC_PHYS_PAGES") * os.sysconf("SC_PAGE_SIZE")
    t_gen = time.time() - t_gen
    t0 = time.time()
    out, ctl = B.compress_device(buf.data_ptr(), n, level=7, threads=os.cpu_count(), processors=os.cpu_count(), ramsize=ram, window=328,
                                 host_threads=16, gpu_slots=8, copy=False)
    dt = time.time() - t0
    _note("cfg5: 32 GiB random (torch.randint on the GPU, seed 5), -L7 -w 328, one chunk, input in HBM: %.1f s = %.1f MB/s, image %d bytes "
          "(every literal block stored through the lz4 gate), cold pools; generating the input took %.1f s" % (dt, (n >> 20) / dt, len(out), t_gen))
    assert len(out) > n  # incompressible: stored blocks + headers
    # Incompressible input: no match, every literal block stored -- the image IS the input cut into stored blocks.
    # Walk the container (every header of the chain), check that, and compare block payloads with the source where
    # they stand; the full 32 GiB decode (CRC + MD5 over everything) is left to tools/full_configs.sh: two minutes.
    import lrz_decode
    view = out.view()
    hdr, chunks = lrz_decode.parse(view)
    assert hdr["st_size"] == n and len(chunks) == 1 and bytes(view[-16:]) == bytes(ctl.hash_resblock)
    lit = chunks[0]["streams"][1]  # (c_type, offset, c_len, u_len)
    assert sum(b[3] for b in lit) == n and all(b[0] == 3 and b[2] == b[3] for b in lit)
    pos = 0
    for k, (c_type, off, c_len, u_len) in enumerate(lit):
        if k in (0, 1, len(lit) // 2, len(lit) - 1):
            m = min(c_len, 64 << 20)
            assert torch.equal(torch.frombuffer(bytearray(view[off:off + m]), dtype=torch.uint8), buf[pos:pos + m].cpu())
        pos += u_len


def test_cfg4_full_10gib_zstd_round_trip(B):
    """cfg 4 at FULL size (BASELINE configs[3]): a 10 GiB tar of 40 copies of a 256 MiB synthetic source tree (a sixteenth of it generated line by line, the rest
    are alphabet permutations of those files), --zstd
    --zstd-level 15 -w 26 (=> rzip level 6, 4 chunks, zstd blocks through the host libzstd like the reference), host
    input; the image through the library's decoder gives the input back (every chunk CRC, the MD5 over all of it).  The
    byte-for-byte comparison with the oracle is test_cfg4_source_tree_tar_zstd15's, at the size the oracle finishes in
    seconds."""
    import time
    t_gen = time.time()
    data = datagen.source_tree_tar(40, 256 << 20, seed=7, variants=16)
    t_gen = time.time() - t_gen
    assert len(data) >= 10 << 30
    ram = os.sysconf("SC_PHYS_PAGES") * os.sysconf("SC_PAGE_SIZE")
    ncpu = os.cpu_count()
    t0 = time.time()
    got, ctl = B.compress_buffer(data, level=7, threads=ncpu, processors=ncpu, ramsize=ram, window=26, zstd=True, zstd_level=15)
    dt = time.time() - t0
    _note("cfg4: %d MiB tar of 40 copies of a 256 MiB synthetic source tree, --zstd --zstd-level 15 -w 26 (rzip level 6), host input: %.1f s = %.1f MB/s, "
          "image %d bytes; generating the input took %.1f s" % (len(data) >> 20, dt, (len(data) >> 20) / dt, len(got), t_gen))
    info = B.file_info(got)
    assert info.chunks == -(-len(data) // (26 * 104857600)) and info.st_size == len(data)
    assert got[17] == (6 << 4) + 4 and got[18] == 15 and got[19] >> 4 == 6
    t0 = time.time()
    back = B.decompress_buffer(got)
    assert len(back) == len(data) and back == data
    _note("cfg4: round trip ok (%.1f s)" % (time.time() - t0))


def test_block_beyond_the_device_is_refused_at_once(B):
    """The reference sizes an LZMA block as max(limit, overhead - dict) / threads (src/stream.c:1316-1323): with -p1 a
    3 GiB file is ONE 3 GiB block.  The GPU match finder keeps ~142 B per block byte resident, so a block has a
    ceiling on a given device (lrzgpu_max_block_bytes: ~1.9 GB on 288 GB); a plan above it is refused with
    LRZGPU_E_BLOCK_TOO_LARGE before anything is scanned -- not with an out-of-memory error minutes into the run."""
    import torch
    L = B.lib()
    L.lrzgpu_max_block_bytes.restype = C.c_int64
    L.lrzgpu_max_block_bytes.argtypes = [C.c_int]
    ceiling = L.lrzgpu_max_block_bytes(0)
    total = torch.cuda.get_device_properties(0).total_memory
    assert (total - (12 << 30)) // 160 < ceiling < total // 130, (ceiling, total)
    if total >= 250 << 30:
        assert ceiling > int(1.2 * (1 << 30))  # the block of the reference's defaults on a 64 GB, 8-core host and a multi-GB file
    n = 3 << 30
    ram = 64 << 30
    plan, _ = B.plan(n, level=7, threads=1, processors=1, ramsize=ram)
    assert plan.stream_bufsize == n > ceiling
    buf = torch.zeros(n + 256, dtype=torch.uint8, device="cuda")
    with pytest.raises(RuntimeError, match="rc=-108"):  # (that it comes at once: tests/test_zz_perf_gpu.py)
        B.compress_device(buf.data_ptr(), n, level=7, threads=1, processors=1, ramsize=ram, host_threads=4)
