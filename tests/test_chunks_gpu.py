"""Multi-chunk files: chunks scanned concurrently on one GPU, the victim_round chain across chunks
(src/rzip.c:308), chunk-sharded compression for one-process-per-GPU runs, bounded-memory fd path.
Everything is compared byte for byte with the oracle driver."""
import ctypes as C
import hashlib
import importlib.util
import os

import pytest

import datagen

pytestmark = pytest.mark.gpu

RAM = 80 * 100 << 20


def _bench():
    spec = importlib.util.spec_from_file_location("lrz_bench_prof2", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    return bench


def _profile(B, bench):
    p = bench.Profile()
    B.lib().lrzgpu_profile_get.argtypes = None  # other test modules load their own copy of bench.Profile
    B.lib().lrzgpu_profile_get(C.byref(p))
    return p


@pytest.fixture(scope="module")
def cfg3_small():
    """BASELINE config 3 at reduced size with the SAME chunk count: 8 chunks (-w 1: 7 x 100 MiB + 60 MiB),
    25 MiB base block so that every chunk holds internal long-range redundancy (four copies per chunk: the oracle's
    single-threaded scan, most of this test's time, runs over matches three quarters of the way)."""
    return datagen.cfg3(7 * 104857600 + 60 * 1048576, 25 * 1048576, seed=3)


def test_cfg3_reduced_concurrent_chunks(B, O, cfg3_small):
    data = cfg3_small
    want, fs = O.compress_buffer(data, compression_level=7, threads=8, processors=16, ramsize=RAM, window=1, workers=16)
    assert fs.n_chunks == 8
    got, ctl = B.compress_buffer(data, level=7, threads=8, processors=16, ramsize=RAM, window=1, host_threads=16, gpu_slots=6)
    assert ctl.stream_bufsize == fs.stream_bufsize
    assert bytes(ctl.hash_resblock) == hashlib.md5(data).digest()
    assert got == want
    # (one scanner -- the chunks strictly one after the other -- is test_victim_round_chain_rescans' configuration)


_VICTIM_DATA = []


def _moving_victim_data(O):
    """Three -w1 chunks whose victim_round does not come back to 0 (checked with the oracle's own scan; made once per run:
    four tests use it)."""
    if not _VICTIM_DATA:
        data = datagen.victim_mover(2 * 104857600 + 20 * 1048576, O.hash_index(), seed=3, every=32768)
        vr = O.rzip_chunk(data[:104857600], level=7)[4]
        assert vr != 0, "generator no longer moves victim_round: pick another seed"
        _VICTIM_DATA.append(data)
    return _VICTIM_DATA[0]


def test_victim_round_chain_rescans(B, O):
    """Permutations of one 31-byte multiset share a tag without ever matching: they pile up in the table until
    insert_hash()'s static victim_round moves (src/rzip.c:308-343).  A chunk that was scanned ahead of its
    predecessor with the predicted value 0 then has to be scanned again; same bytes as the serial chain."""
    bench = _bench()
    data = _moving_victim_data(O)
    want, fs = O.compress_buffer(data, compression_level=7, threads=4, processors=8, ramsize=RAM, window=1, workers=16)
    assert fs.n_chunks == 3
    B.lib().lrzgpu_profile_reset()
    got, _ = B.compress_buffer(data, level=7, threads=4, processors=8, ramsize=RAM, window=1, host_threads=16, scan_slots=3)
    assert got == want
    p = _profile(B, bench)
    assert p.victim_rescans >= 1
    B.lib().lrzgpu_profile_reset()
    got, _ = B.compress_buffer(data, level=7, threads=4, processors=8, ramsize=RAM, window=1, host_threads=16, scan_slots=1)
    assert got == want
    assert _profile(B, bench).victim_rescans == 0


def _sharded_c_abi(B, data, world, **kw):
    """lrzgpu_compress_sharded (csrc/shard.cpp) with every rank a thread of this process and an in-process transport
    (queues for send / recv, a barrier-protected sum for the all-reduce): the protocol, the library's own GPU
    compressor per rank and the hand-off run exactly as under torch.distributed, on one GPU."""
    import queue
    import threading
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("lrz_sharded_t", os.path.join(root, "lrzip-next_amd", "sharded.py"))
    SH = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(SH)
    L = B.lib()
    L.lrzgpu_compress_sharded.argtypes = [C.c_void_p, C.c_char_p, C.c_int64, C.c_void_p, C.POINTER(C.POINTER(C.c_ubyte)),
                                          C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    chan = {(a, b): queue.Queue() for a in range(world) for b in range(world)}
    bar = threading.Barrier(world)
    acc, lock = {}, threading.Lock()

    def make(rank):
        def allreduce(_c, vals, count):
            gen = allreduce.gen = getattr(allreduce, "gen", 0) + 1
            mine = [vals[i] for i in range(count)]
            with lock:
                tot = acc.setdefault(gen, [0] * count)
                for i in range(count):
                    tot[i] += mine[i]
            bar.wait()
            for i in range(count):
                vals[i] = acc[gen][i]
            bar.wait()
            return 0

        def send(_c, dst, buf, n):
            chan[(rank, dst)].put(C.string_at(buf, n))
            return 0

        def recv(_c, src, buf, n):
            b = chan[(src, rank)].get(timeout=600)
            assert len(b) == n
            C.memmove(buf, b, n)
            return 0

        cbs = (SH.ALLREDUCE(allreduce), SH.SEND(send), SH.RECV(recv))
        return SH.ShardComm(None, rank, world, *cbs), cbs

    res = [None] * world

    def worker(rank):
        comm, keep = make(rank)
        c = B.make_control(host_threads=max(2, 8 // world), **kw)
        out = C.POINTER(C.c_ubyte)()
        olen, redone = C.c_int64(), C.c_int64()
        rc = L.lrzgpu_compress_sharded(C.byref(c), data, len(data), C.byref(comm), C.byref(out), C.byref(olen), C.byref(redone))
        img = C.string_at(out, olen.value) if (rc == 0 and out) else None
        if out:
            C.CDLL(None).free(out)
        res[rank] = (rc, img, redone.value)

    th = [threading.Thread(target=worker, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(900)
    assert all(r is not None and r[0] == 0 for r in res), res
    assert all(r[1] is None for r in res[1:])
    return res[0][1], res[0][2]


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_c_abi_entry_equals_single_process(B, O, world):
    # (-L1 and a 30 MiB -m: 20 MiB chunks, 5 MiB blocks -- five chunks over 2 or 3 ranks without a 400 MB input)
    kw = dict(level=1, threads=2, processors=2, ramsize=30 * 1048576)
    data = datagen.cfg3(90 * 1048576 + 12345, 12 * 1048576, seed=5)
    want, fs = O.compress_buffer(data, compression_level=1, threads=2, processors=2, ramsize=30 * 1048576, workers=16)
    assert fs.n_chunks == 5
    got, redone = _sharded_c_abi(B, data, world, **kw)
    assert got == want  # (rzip level 1: victim_round moves in every chunk, the chain is put right by redoing -- `redone` > 0)


def test_sharded_c_abi_entry_redoes_wrong_guesses(B, O):
    kw = dict(level=7, threads=4, processors=8, ramsize=RAM, window=1)
    data = _moving_victim_data(O)
    want, _ = O.compress_buffer(data, compression_level=7, threads=4, processors=8, ramsize=RAM, window=1, workers=16)
    got, redone = _sharded_c_abi(B, data, 3, **kw)
    assert got == want and redone >= 1


def test_fd_path_streams_chunks(B, O, tmp_path):
    """Regular files are read chunk by chunk and written chunk by chunk (src/rzip.c:1057-1107,
    src/stream.c:1772-1821): same bytes as the memory-to-memory call, at an output offset too."""
    data = datagen.cfg3(104857600 + 45 * 1048576 + 999, 15 * 1048576, seed=9)
    want, _ = O.compress_buffer(data, compression_level=7, threads=4, processors=8, ramsize=RAM, window=1, workers=16)
    src = tmp_path / "in.bin"
    src.write_bytes(data)
    B.compress_file(str(src), str(tmp_path / "a.lrz"), level=7, threads=4, processors=8, ramsize=RAM, window=1, host_threads=16)
    assert (tmp_path / "a.lrz").read_bytes() == want
    B.compress_file(str(src), str(tmp_path / "b.part"), rzip_only_fd=True, level=7, threads=4, processors=8, ramsize=RAM,
                    window=1, host_threads=16)
    assert (tmp_path / "b.part").read_bytes() == want[21:]


def test_file_whose_scan_outruns_the_hash(B, tmp_path):
    """A file of many chunks whose scans take no time (zeros: one long match per chunk, compared at HBM speed) against
    the whole-input MD5's ~1 GB/s on one host thread: the committer lets go of chunk after chunk while the hash is
    still reading them from their copies in HBM.  Those copies are bounded (readers wait while more than two are held
    for the hash alone: csrc/scan_run.cpp) -- the image is the memory-to-memory call's, which
    tests/test_compress_gpu.py pins to the oracle's."""
    n = 7 * 104857600 + 4321  # (two scanners: the readers may be three chunks ahead of the committer + two held for the hash)
    data = bytes(n)
    kw = dict(level=7, threads=4, processors=8, ramsize=RAM, window=1, host_threads=8, scan_slots=2, no_compress=True)  # (-n: nothing but the scan)
    want, _ = B.compress_buffer(data, **kw)
    src = tmp_path / "zeros.bin"
    src.write_bytes(data)
    B.compress_file(str(src), str(tmp_path / "z.lrz"), **kw)
    got = (tmp_path / "z.lrz").read_bytes()
    assert got == want
    assert B.file_info(got).chunks == 8
