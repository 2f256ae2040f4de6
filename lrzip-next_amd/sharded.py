"""One file across one process per GPU: chunk k of the file belongs to rank k mod world.

rzip chunks are independent units of the .lrz format (own header, hash table, CRC, block offsets relative to the
chunk; src/rzip.c:599-626, src/stream.c:1740-1770).  Every rank runs the whole path for its own chunks
(lrzgpu_compress_chunks*) and ends up with finished chunk images; the only things that travel are
  * three integers per chunk (victim_round in / out, image length): one all_reduce,
  * the chunk images themselves, to rank 0, in file order: point-to-point send/recv -- the chunk hand-off
    (RCCL over xGMI with backend "nccl", TCP with "gloo" in the CPU tests).
The one value that crosses a chunk boundary in the reference is insert_hash()'s static victim_round
(src/rzip.c:308): ranks start their chunks from a prediction, rank 0's view of the (in, out) table shows which
chunk (if any) started from the wrong value, its owner redoes that one chunk, until the chain is consistent.
Rank 0 lays out magic + chunks + MD5 (lrzgpu_assemble_chunks).  Orchestration only: no byte of the data path is
computed here.
"""


def owner(k, world):
    return k % world


def compress_sharded(compress_fn, n_chunks, rank, world, dist, torch, device, max_rounds=None):
    """compress_fn(first, stride, victim_in) -> {chunk index: (victim_in, victim_out, image bytes)}: the chunks
    k % stride == first of the file, each started from victim_in[k] where that is >= 0.
    Returns (images in file order on rank 0 else None, number of chunks redone over all ranks)."""
    images, chain = {}, {}
    if rank < n_chunks:
        for k, (vin, vout, img) in compress_fn(rank, world, None).items():
            images[k], chain[k] = img, (vin, vout)
    redone = 0
    rounds = 0
    while True:
        meta = torch.zeros((n_chunks, 3), dtype=torch.int64, device=device)
        for k, (vin, vout) in chain.items():
            meta[k, 0], meta[k, 1], meta[k, 2] = vin, vout, len(images[k])
        if world > 1:
            dist.all_reduce(meta, op=dist.ReduceOp.SUM)
        m = meta.cpu().tolist()
        # the chain of src/rzip.c:308: chunk k must have started from what chunk k-1 left
        bad = [k for k in range(1, n_chunks) if m[k][0] != m[k - 1][1]]
        if not bad:
            break
        k0 = bad[0]  # only the first wrong chunk is certain to be wrong: its new end value decides about the rest
        if owner(k0, world) == rank:
            victim = [-1] * n_chunks
            victim[k0] = m[k0 - 1][1]
            vin, vout, img = compress_fn(k0, max(n_chunks, k0 + 1), victim)[k0]  # chunk k0 alone
            images[k0], chain[k0] = img, (vin, vout)
        redone += 1
        rounds += 1
        if max_rounds is not None and rounds > max_rounds:
            raise RuntimeError("victim_round chain did not converge")
    if world == 1:
        return [images[k] for k in range(n_chunks)], redone
    # chunk hand-off to rank 0, in file order
    if rank == 0:
        out = []
        for k in range(n_chunks):
            if owner(k, world) == 0:
                out.append(images[k])
            else:
                t = torch.empty(int(m[k][2]), dtype=torch.uint8, device=device)
                dist.recv(t, src=owner(k, world))
                out.append(t.cpu().numpy().tobytes())
        return out, redone
    for k in range(rank, n_chunks, world):
        t = torch.frombuffer(bytearray(images[k]), dtype=torch.uint8).to(device)
        dist.send(t, dst=0)
    return None, redone
