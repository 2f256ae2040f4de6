"""Chunk-per-rank sharding of one file across GPUs (one process per GPU, torch.distributed).

rzip chunks are independent units of the .lrz format (own header, hash table, CRC); the only
state that crosses a chunk boundary in the reference is insert_hash()'s static `victim_round`
(src/rzip.c:308).  Rank r owns chunks r, r+G, ...; every chunk is first scanned with the value its
predecessor is EXPECTED to leave (0), the (in, out) pairs are all-gathered, and only chunks whose
assumption turned out wrong are re-run.  There is no collective on the data path: ranks read their
own byte ranges; the results (tokens, literal streams or finished chunk payloads) are gathered to
rank 0, which writes the file in chunk order.  Backend: "nccl" (= RCCL over xGMI) on GPUs, "gloo"
in the CPU tests.
"""


def shard_chunks(n_chunks, world):
    """chunk indices owned by each rank (round robin, so consecutive chunks run concurrently)."""
    return [list(range(r, n_chunks, world)) for r in range(world)]


def chunk_ranges(st_size, chunk_size):
    """(offset, length) of every rzip chunk of a file (src/rzip.c:1041-1053)."""
    if st_size == 0:
        return [(0, 0)]
    out, off = [], 0
    while off < st_size:
        n = min(chunk_size, st_size - off)
        out.append((off, n))
        off += n
    return out


def run_sharded(chunk_fn, n_chunks, rank=0, world=1, dist=None, initial_victim_round=0, max_rounds=None):
    """Runs chunk_fn(index, victim_round_in) -> (payload, victim_round_out) for this rank's chunks.

    Returns (payloads, stats): on rank 0 `payloads` is the list of all chunk payloads in chunk
    order, elsewhere None.  stats = {"reruns": total number of chunk re-runs over all ranks}.
    """
    mine = shard_chunks(n_chunks, world)[rank]
    assumed = {k: (initial_victim_round if k == 0 else 0) for k in mine}
    results = {}
    todo = list(mine)
    reruns = 0
    rounds = 0
    while True:
        for k in todo:
            payload, vr_out = chunk_fn(k, assumed[k])
            results[k] = (payload, vr_out)
        local = {k: (assumed[k], results[k][1]) for k in mine}
        if world > 1:
            gathered = [None] * world
            dist.all_gather_object(gathered, local)
        else:
            gathered = [local]
        table = {}
        for g in gathered:
            table.update(g)
        # the value each chunk must start from, given what its predecessor actually left
        expected = {0: initial_victim_round}
        for k in range(1, n_chunks):
            expected[k] = table[k - 1][1]
        wrong = [k for k in range(n_chunks) if table[k][0] != expected[k]]
        if not wrong:
            break
        # only the FIRST wrong chunk is certain to be wrong with this `expected`; re-running all of
        # them is still correct (the loop re-validates) and converges in at most n_chunks rounds
        todo = [k for k in wrong if k in assumed]
        for k in todo:
            assumed[k] = expected[k]
        reruns += len(wrong)
        rounds += 1
        if max_rounds is not None and rounds > max_rounds:
            raise RuntimeError("victim_round speculation did not converge")
    mine_payloads = {k: results[k][0] for k in mine}
    if world > 1:
        allp = [None] * world if rank == 0 else None
        dist.gather_object(mine_payloads, allp, dst=0)
    else:
        allp = [mine_payloads]
    if rank != 0:
        return None, {"reruns": reruns}
    merged = {}
    for g in allp:
        merged.update(g)
    return [merged[k] for k in range(n_chunks)], {"reruns": reruns}
