"""One file across one process per GPU -- the TRANSPORT half only (and since round 4 only the fall-back and test
transport: bench.py --gpus N hands the chunks off through the library's own RCCL transport in C, rccl_comm() below).

The protocol (chunk k -> rank k mod world, the victim_round chain check and redo, the chunk hand-off to rank 0, the
layout of the one .lrz) lives in the library behind the C ABI (csrc/shard.cpp: lrzgpu_compress_sharded*,
lrzgpu_shard_protocol).  It asks its caller for three things: sum a few int64 over all ranks, send bytes to a rank,
receive bytes from a rank.  This module is those three callbacks over torch.distributed (backend "nccl" = RCCL over
xGMI on the GPU box: host bytes are staged through a device tensor for the send / recv; "gloo" in the CPU tests),
for bench.py and tests/.  Nothing of the data path is computed here.
"""
import ctypes as C

ALLREDUCE = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_int64), C.c_int)
SEND = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int64)
RECV = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int64)


class ShardComm(C.Structure):
    _fields_ = [("ctx", C.c_void_p), ("rank", C.c_int), ("world", C.c_int), ("allreduce_sum_i64", ALLREDUCE),
                ("send", SEND), ("recv", RECV)]


def owner(k, world):
    return k % world


def torch_comm(rank, world, dist, torch, device, piece=1 << 30):
    """lrzgpu_shard_comm over torch.distributed.  `device`: where send / recv tensors live (the rank's GPU with
    RCCL, cpu with gloo).  Returns (ShardComm, keepalive): keep the second alive as long as the first is used."""
    import numpy as np

    def as_numpy(ptr, n):
        return np.ctypeslib.as_array((C.c_ubyte * n).from_address(ptr))

    def allreduce(_ctx, vals, count):
        try:
            a = np.ctypeslib.as_array(vals, shape=(count,))
            t = torch.from_numpy(a.copy()).to(device)
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            a[:] = t.cpu().numpy()
            return 0
        except Exception:  # a callback must not raise into C
            return -1

    def send(_ctx, dst, buf, n):
        try:
            for o in range(0, n, piece):  # (pieces: one device staging tensor of at most `piece` bytes)
                k = min(piece, n - o)
                t = torch.from_numpy(as_numpy(buf + o, k)).to(device)
                dist.send(t, dst=dst)
            return 0
        except Exception:
            return -1

    def recv(_ctx, src, buf, n):
        try:
            for o in range(0, n, piece):
                k = min(piece, n - o)
                t = torch.empty(k, dtype=torch.uint8, device=device)
                dist.recv(t, src=src)
                as_numpy(buf + o, k)[:] = t.cpu().numpy()
            return 0
        except Exception:
            return -1

    cbs = (ALLREDUCE(allreduce), SEND(send), RECV(recv))
    return ShardComm(None, rank, world, *cbs), cbs


def rccl_comm(lib, rank, world, device, bcast_id):
    """lrzgpu_shard_comm over the library's own RCCL transport (csrc/shard_rccl.cpp): ncclAllReduce / ncclSend /
    ncclRecv in C, no Python on the hand-off.  bcast_id(bytes or None) -> bytes carries rank 0's 128-byte unique id to
    every rank (the caller's bootstrap).  Returns (ShardComm, close); raises RuntimeError when RCCL cannot be used."""
    # every rank takes part in the broadcast whatever it finds locally (a rank that left before it would leave the others
    # waiting in it): all zero = "rank 0 has none"
    have = bool(lib.lrzgpu_rccl_available())
    uid = None
    if rank == 0:
        raw = (C.c_ubyte * 128)()
        rc = lib.lrzgpu_rccl_unique_id(raw) if have else -1
        uid = bytes(raw) if rc == 0 else bytes(128)
    uid = bcast_id(uid)
    if not have:
        raise RuntimeError("no librccl in this process")
    if uid == bytes(128):
        raise RuntimeError("rank 0 could not make a unique id")
    comm = ShardComm()
    rc = lib.lrzgpu_rccl_comm_create((C.c_ubyte * 128).from_buffer_copy(uid), rank, world, device, C.byref(comm))
    if rc:
        raise RuntimeError("lrzgpu_rccl_comm_create rc=%d" % rc)
    return comm, (lambda: lib.lrzgpu_rccl_comm_destroy(C.byref(comm)))
