"""CPU (gloo, world_size 2) test of the one-file-across-ranks path through the C ABI (lrzgpu_shard_protocol,
csrc/shard.cpp): chunk ownership, the victim_round chain check and redo, the chunk hand-off to rank 0 and the layout
of the one file, over the three transport callbacks of lrzip-next_amd/sharded.py on torch.distributed.
There is no GPU in this container, so the per-rank chunk compressor handed to the protocol is built from the ORACLE
scan + the product's host-only container layout; on the GPU box bench.py calls lrzgpu_compress_sharded_dev, where the
compressor is the library's own GPU path (tests/test_chunks_gpu.py plays all ranks in one process against that)."""
import hashlib
import importlib.util
import os
import socket
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _load(name, rel):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, rel))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


RAM = (2 << 20) * 3 // 2  # max_chunk = ramsize/3*2 = 2 MiB


def _data():
    sys.path.insert(0, HERE)
    import datagen
    import oracle_lib as O
    # 6 chunks; permutations of one multiset make victim_round really move between chunks
    return datagen.long_range(3 * 1048576 + 123, seed=2, base_frac=0.2) + datagen.victim_mover(8 << 20, O.hash_index(), seed=5, every=4096)[:8 << 20]


def _chunk_image(B, O, data, ranges, k, victim_in):
    """One chunk's bytes as they stand in the -n file: oracle scan from victim_in + the product's stored-block
    layout of that one chunk, chunk header patched to this chunk's place in the file (eof flag)."""
    off, n = ranges[k]
    s0, s1, st, crc, vr_out = O.rzip_chunk(data[off:off + n], level=7, chunk_bytes=B.chunk_bytes_for(n), victim_round=victim_in)
    one = bytearray(B.container_store(n, [n], [s0], [s1], bytes(16), no_compress=True, threads=1, ramsize=RAM)[21:-16])
    one[1] = 1 if k + 1 == len(ranges) else 0
    return vr_out, bytes(one)


def _ranges(n, chunk):
    out, off = [], 0
    while off < n:
        out.append((off, min(chunk, n - off)))
        off += chunk
    return out or [(0, 0)]


def _worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, HERE)
    import oracle_lib as O
    SH = _load("lrzip_next_amd_sharded", "lrzip-next_amd/sharded.py")
    B = _load("lrzip_next_amd_bindings", "lrzip-next_amd/bindings.py")
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        data = _data()
        ctl, chunk = B.plan(len(data), no_compress=True, threads=1, ramsize=RAM)
        ranges = _ranges(len(data), chunk)
        calls = []

        def compress_fn(first, stride, victim_in):
            got = {}
            for k in range(first, len(ranges), stride):
                vin = victim_in[k] if victim_in is not None and victim_in[k] >= 0 else 0
                vout, img = _chunk_image(B, O, data, ranges, k, vin)
                calls.append((k, vin, vout))
                got[k] = (vin, vout, img)
            return got

        comm, keep = SH.torch_comm(rank, world, dist, torch, torch.device("cpu"))
        got, redone = B.shard_protocol(len(data), comm, compress_fn, hashlib.md5(data).digest(), no_compress=True, threads=1, ramsize=RAM)
        if rank == 0:
            want, fs = O.compress_buffer(data, no_compress=1, threads=1, ramsize=RAM)
            q.put(("result", got == want, len(ranges), fs.n_chunks, redone))
        else:
            assert got is None
        # where this rank's seconds went (lrzgpu_profile.shard_s, what bench.py --gpus N prints as critical_path)
        import ctypes as C
        bench = _load("lrz_bench_mod", "bench.py")
        prof = bench.Profile()
        B.lib().lrzgpu_profile_get.argtypes = [C.POINTER(bench.Profile)]
        B.lib().lrzgpu_profile_get(C.byref(prof))
        q.put(("calls", rank, calls, list(prof.shard_s)))
    finally:
        dist.destroy_process_group()


def test_two_ranks_chunk_sharding_and_victim_round_protocol():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    msgs = [q.get(timeout=300) for _ in range(3)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    res = [m for m in msgs if m[0] == "result"][0]
    assert res[1], "sharded .lrz differs from the single-process oracle"
    assert res[2] == res[3] >= 5
    assert res[4] >= 1, "the data was meant to break the victim_round prediction at least once"
    calls = {m[1]: m[2] for m in msgs if m[0] == "calls"}
    stage_s = {m[1]: m[3] for m in msgs if m[0] == "calls"}
    for r in (0, 1):  # own chunks, chain check, redone, hand-off, (hash), protocol wall -- seconds of that rank
        own, check, redo, handoff, _, wall = stage_s[r]
        assert wall > 0 and own > 0 and handoff >= 0 and wall >= own and wall + 1e-6 >= own + check + redo + handoff
    assert max(stage_s[0][2], stage_s[1][2]) > 0  # somebody redid a chunk, and it shows
    k0 = {c[0] for c in calls[0]}
    k1 = {c[0] for c in calls[1]}
    assert k0 and k1 and not (k0 & k1)  # both ranks did work, on disjoint chunk sets


def test_single_rank_chain_redo():
    """world 1, no transport: a forced wrong prediction is redone until the chain is consistent (the C protocol
    over a toy compressor whose chunk 1 leaves victim_round 3)."""
    sys.path.insert(0, HERE)
    SH = _load("lrzip_next_amd_sharded", "lrzip-next_amd/sharded.py")
    B = _load("lrzip_next_amd_bindings", "lrzip-next_amd/bindings.py")
    seen = []
    ctl, chunk = B.plan(4 * (2 << 20), no_compress=True, threads=1, ramsize=RAM)
    assert chunk == 2 << 20

    def compress_fn(first, stride, victim_in):
        got = {}
        for k in range(first, 4, stride):
            vin = victim_in[k] if victim_in is not None and victim_in[k] >= 0 else 0
            seen.append((k, vin))
            got[k] = (vin, 3 if k == 1 else vin, b"p%d@%d" % (k, vin))
        return got

    comm = SH.ShardComm(None, 0, 1)
    img, redone = B.shard_protocol(4 * (2 << 20), comm, compress_fn, bytes(16), no_compress=True, threads=1, ramsize=RAM)
    assert img[21:-16] == b"p0@0" + b"p1@0" + b"p2@3" + b"p3@3" and redone == 2
    assert seen == [(0, 0), (1, 0), (2, 0), (3, 0), (2, 3), (3, 3)]
    assert SH.owner(5, 2) == 1


def _failing_worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, HERE)
    SH = _load("lrzip_next_amd_sharded", "lrzip-next_amd/sharded.py")
    B = _load("lrzip_next_amd_bindings", "lrzip-next_amd/bindings.py")
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        n = 4 * (2 << 20)

        def compress_fn(first, stride, victim_in):
            if rank == 1:
                raise MemoryError("rank 1 cannot compress its chunks")
            return {k: (0, 0, b"p%d" % k) for k in range(first, 4, stride)}

        comm, keep = SH.torch_comm(rank, world, dist, torch, torch.device("cpu"))
        try:
            B.shard_protocol(n, comm, compress_fn, bytes(16), no_compress=True, threads=1, ramsize=RAM)
            q.put((rank, "no error"))
        except RuntimeError as e:
            q.put((rank, str(e)))
    finally:
        dist.destroy_process_group()


def test_a_failing_rank_fails_every_rank_instead_of_hanging_them():
    """a rank whose compressor fails used to return at once and leave its peers blocked in the all-reduce.
    Now the failure travels through the collective: the failing rank returns its own error, the other LRZGPU_E_PEER
    (-107), before any send / recv."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_failing_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    msgs = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert "rc=-107" in msgs[0], msgs
    assert "rc=-1" in msgs[1] and "rc=-107" not in msgs[1], msgs
