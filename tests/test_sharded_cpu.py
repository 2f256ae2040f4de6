"""CPU (gloo, world_size 2) test of the chunk-per-rank path: sharding, the victim_round speculation
protocol, gather to rank 0 and container assembly.  The chunk scanner plugged in here is the ORACLE
(no GPU in this container); on the GPU box the same orchestration drives lrzgpu_hash_search."""
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


def _worker(rank, world, port, q):
    import torch.distributed as dist
    sys.path.insert(0, HERE)
    import datagen
    import oracle_lib as O
    P = _load("lrz_parallel", "lrzip-next_amd/parallel.py")
    B = _load("lrzip_next_amd_bindings", "lrzip-next_amd/bindings.py")
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        # 5 chunks; a pattern that drives the round-robin victim path so victim_round really moves
        pat = (b"0123456789abcdefghijklmnopqrstu" * 40 + b"XYZ") * 900
        data = datagen.long_range(5 * 1048576 + 123, seed=2, base_frac=0.2) + pat
        ram = (2 << 20) * 3 // 2  # max_chunk = ramsize/3*2 = 2 MiB
        ctl, chunk = B.plan(len(data), no_compress=True, threads=1, ramsize=ram)
        ranges = P.chunk_ranges(len(data), chunk)
        calls = []

        def chunk_fn(k, vr_in):
            off, n = ranges[k]
            s0, s1, st, crc, vr_out = O.rzip_chunk(data[off:off + n], level=7, chunk_bytes=B.chunk_bytes_for(n),
                                                   victim_round=vr_in)
            calls.append((k, vr_in, vr_out))
            return (s0, s1), vr_out

        payloads, stats = P.run_sharded(chunk_fn, len(ranges), rank, world, dist)
        if rank == 0:
            got = B.container_store(len(data), [n for _, n in ranges], [p[0] for p in payloads],
                                    [p[1] for p in payloads], hashlib.md5(data).digest(), no_compress=True,
                                    threads=1, ramsize=ram)
            want, fs = O.compress_buffer(data, no_compress=1, threads=1, ramsize=ram)
            q.put(("result", got == want, len(ranges), fs.n_chunks, stats["reruns"]))
        q.put(("calls", rank, calls))
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
    assert res[2] == res[3] >= 3
    calls = {m[1]: m[2] for m in msgs if m[0] == "calls"}
    # both ranks did work, on disjoint chunk sets
    k0 = {c[0] for c in calls[0]}
    k1 = {c[0] for c in calls[1]}
    assert k0 and k1 and not (k0 & k1)


def test_shard_helpers():
    P = _load("lrz_parallel", "lrzip-next_amd/parallel.py")
    assert P.shard_chunks(5, 2) == [[0, 2, 4], [1, 3]]
    assert P.shard_chunks(3, 8)[3:] == [[]] * 5
    assert P.chunk_ranges(0, 100) == [(0, 0)]
    assert P.chunk_ranges(250, 100) == [(0, 100), (100, 100), (200, 50)]
    # single-process path with a forced mis-speculation: chunk 1 leaves victim_round 3
    seen = []

    def fn(k, vr):
        seen.append((k, vr))
        return ("p%d@%d" % (k, vr), 3 if k == 1 else vr)

    payloads, stats = P.run_sharded(fn, 4)
    assert payloads == ["p0@0", "p1@0", "p2@3", "p3@3"] and stats["reruns"] >= 1
