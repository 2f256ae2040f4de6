"""The chunk hand-off's transport in C over RCCL (csrc/shard_rccl.cpp; include/lrzgpu.h lrzgpu_rccl_*).  One GPU here, so
a world of one: communicator bootstrap from a unique id, the all-reduce callback, the staging / ncclSend / ncclRecv
path as a grouped send + receive to the rank itself (lrzgpu_rccl_loopback: several 32 MiB pieces and a ragged last one),
and the sharded entry point driven with this communicator.  The N-rank protocol is covered on the CPU over gloo
(tests/test_sharded_cpu.py) and with ranks as threads on the GPU (tests/test_chunks_gpu.py)."""
import ctypes as C
import importlib.util
import os
import sys

import numpy as np
import pytest

import datagen
from test_compress_gpu import RAM

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _sharded_module():
    name = "lrzip_next_amd_sharded"
    if name in sys.modules:
        return sys.modules[name]
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "lrzip-next_amd", "sharded.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


@pytest.fixture(scope="module")
def comm(B):
    SH = _sharded_module()
    L = B.lib()
    if not L.lrzgpu_rccl_available():
        pytest.fail("librccl is part of the ROCm image: the C transport must find it")
    c, close = SH.rccl_comm(L, 0, 1, 0, lambda uid: uid)
    yield c
    assert close() == 0


def test_allreduce_and_loopback(B, comm):
    L = B.lib()
    vals = (C.c_int64 * 7)(1, -2, 3, 1 << 40, 0, -(1 << 50), 7)
    assert comm.allreduce_sum_i64(comm.ctx, vals, 7) == 0
    assert list(vals) == [1, -2, 3, 1 << 40, 0, -(1 << 50), 7]
    for n in (0, 1, 4097, (32 << 20), (70 << 20) + 3):
        src = np.frombuffer(datagen.random_bytes(max(n, 1), seed=n % 1000), dtype=np.uint8)[:n].copy()
        dst = np.zeros(n, dtype=np.uint8)
        rc = L.lrzgpu_rccl_loopback(C.byref(comm), src.ctypes.data_as(C.c_void_p), dst.ctypes.data_as(C.c_void_p), C.c_int64(n))
        assert rc == 0
        assert np.array_equal(src, dst), n
    # the protocol never sends to its own rank; the callbacks refuse it instead of dead-locking
    assert comm.send(comm.ctx, 0, None, 0) != 0 and comm.recv(comm.ctx, 0, None, 0) != 0


def test_sharded_entry_point_with_the_c_transport(B, O, comm):
    import torch
    data = datagen.long_range(250 * 1048576 + 4097, seed=14, base_frac=0.08, mutate_every=300007)
    want, fs = O.compress_buffer(data, compression_level=7, threads=4, processors=8, ramsize=RAM, window=1, workers=8)
    assert fs.n_chunks == 3
    t = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
    out, ctl, redone = B.compress_sharded_dev(t.data_ptr(), t.numel(), comm, level=7, threads=4, processors=8, ramsize=RAM, window=1,
                                              host_threads=8)
    assert out.tobytes() == want and redone == 0
