"""Run by tests/test_shard_rccl_two_ranks_gpu.py in a fresh process: two communicator ranks of csrc/shard_rccl.cpp as
two THREADS on one GPU, over tests/stubs/nccl_stub.cpp (argv[1] = the built stub, argv[2] = scenario)."""
import ctypes as C
import importlib.util
import os
import sys
import threading

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import datagen
from conftest import load_bindings

B = load_bindings()
L = B.lib()
assert L.lrzgpu_rccl_use_library(sys.argv[1].encode()) == 0, "the stand-in could not be loaded"
spec = importlib.util.spec_from_file_location("sharded", os.path.join(ROOT, "lrzip-next_amd", "sharded.py"))
SH = importlib.util.module_from_spec(spec)
spec.loader.exec_module(SH)
scenario = sys.argv[2]

uid_box, uid_ready = {}, threading.Event()


def bcast(uid):
    if uid is not None:
        uid_box["id"] = uid
        uid_ready.set()
    uid_ready.wait(30)
    return uid_box["id"]


def ranks(fn):
    """fn(rank, comm) on two threads, each with its own communicator; returns their results (exceptions re-raised)."""
    out, err = [None, None], [None, None]

    def body(r):
        try:
            comm, close = SH.rccl_comm(L, r, 2, 0, bcast)
            try:
                out[r] = fn(r, comm)
            finally:
                close()
        except BaseException as e:  # noqa
            err[r] = e
    th = [threading.Thread(target=body, args=(r,)) for r in (0, 1)]
    for t in th:
        t.start()
    for t in th:
        t.join(180)
    if any(e is not None for e in err):
        raise RuntimeError("rank errors: %r" % (err,))
    return out


if scenario == "exchange":
    sizes = [0, 1, 4097, 32 << 20, (32 << 20) + 1, (70 << 20) + 3]
    msgs = {n: np.frombuffer(datagen.random_bytes(max(n, 1), seed=n % 997), dtype=np.uint8)[:n].copy() for n in sizes}

    def fn(r, comm):
        vals = (C.c_int64 * 5)(r + 1, -7 * (r + 1), 1 << (40 + r), 0, 5)
        assert comm.allreduce_sum_i64(comm.ctx, vals, 5) == 0
        assert list(vals) == [3, -21, (1 << 40) + (1 << 41), 0, 10], list(vals)
        for n in sizes:  # rank 1 -> rank 0, then back, piece by piece (32 MiB pieces, a ragged last one)
            if r == 1:
                assert comm.send(comm.ctx, 0, msgs[n].ctypes.data_as(C.c_void_p), n) == 0
                back = np.zeros(n, dtype=np.uint8)
                assert comm.recv(comm.ctx, 0, back.ctypes.data_as(C.c_void_p), n) == 0
                assert np.array_equal(back, msgs[n][::-1]), n
            else:
                got = np.zeros(n, dtype=np.uint8)
                assert comm.recv(comm.ctx, 1, got.ctypes.data_as(C.c_void_p), n) == 0
                assert np.array_equal(got, msgs[n]), n
                rev = got[::-1].copy()
                assert comm.send(comm.ctx, 1, rev.ctypes.data_as(C.c_void_p), n) == 0
        # a second all-reduce after the traffic: the streams are in order
        v2 = (C.c_int64 * 1)(10 + r)
        assert comm.allreduce_sum_i64(comm.ctx, v2, 1) == 0 and v2[0] == 21
        return True
    assert ranks(fn) == [True, True]
elif scenario == "sharded":
    import oracle_lib as O
    import torch
    from test_compress_gpu import RAM
    data = datagen.long_range(250 * 1048576 + 4097, seed=14, base_frac=0.08, mutate_every=300007)
    want, fs = O.compress_buffer(data, compression_level=7, threads=4, processors=8, ramsize=RAM, window=1, workers=8)
    assert fs.n_chunks == 3
    t = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()

    def fn(r, comm):
        out, ctl, redone = B.compress_sharded_dev(t.data_ptr(), t.numel(), comm, level=7, threads=4, processors=8, ramsize=RAM, window=1,
                                                  host_threads=4)
        return out.tobytes() if r == 0 else (0 if out is None else len(out))
    res = ranks(fn)
    assert res[0] == want and res[1] == 0, (len(res[0]), res[1])  # rank 0 holds the file, rank 1 nothing
elif scenario == "peer_fails":
    # the sender's third ncclSend fails (NCCL_STUB_FAIL_SEND=3, set by the test): its callback reports the error and
    # aborts the communicator -- which ends the receiver's pending receive instead of leaving it to its ten-minute patience
    n = 150 << 20
    payload = np.zeros(n, dtype=np.uint8)

    def fn(r, comm):
        if r == 1:
            return comm.send(comm.ctx, 0, payload.ctypes.data_as(C.c_void_p), n)
        got = np.zeros(n, dtype=np.uint8)
        return comm.recv(comm.ctx, 1, got.ctypes.data_as(C.c_void_p), n)
    import time
    t0 = time.time()
    res = ranks(fn)
    assert res[0] != 0 and res[1] != 0, res
    assert time.time() - t0 < 60
else:
    raise SystemExit("unknown scenario")
print("TWO-RANKS-OK " + scenario)
