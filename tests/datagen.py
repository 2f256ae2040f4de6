"""Seeded synthetic inputs shared by the tests and bench.py (numpy, fast)."""
import numpy as np


def text_like(n, seed=1, nwords=5000):
    """Word-list pseudo text (LZMA-compressible, few long repeats)."""
    rng = np.random.default_rng(seed)
    lens = rng.integers(2, 10, size=nwords)
    words = [bytes(rng.integers(97, 123, size=int(l), dtype=np.uint8)) + b" " for l in lens]
    out = bytearray()
    # draw in batches
    while len(out) < n:
        idx = rng.integers(0, nwords, size=65536)
        out += b"".join(words[i] for i in idx)
    return bytes(out[:n])


def random_bytes(n, seed=2):
    return np.random.default_rng(seed).integers(0, 256, size=n, dtype=np.uint8).tobytes()


def few_symbols(n, seed=3, k=4):
    return (np.random.default_rng(seed).integers(0, k, size=n, dtype=np.uint8) + 97).astype(np.uint8).tobytes()


def phrase_mix(n, seed=4):
    rng = np.random.default_rng(seed)
    phrases = [rng.integers(0, 256, size=int(rng.integers(3, 41)), dtype=np.uint8).tobytes() for _ in range(50)]
    out = bytearray()
    while len(out) < n:
        idx = rng.integers(0, 50, size=4096)
        out += b"".join(phrases[i] for i in idx)
    return bytes(out[:n])


def sparse_repeats(n, seed=5):
    rng = np.random.default_rng(seed)
    a = rng.integers(0, 256, size=n, dtype=np.uint8)
    if n > 16:
        for _ in range(n // 500 + 1):
            s, d = int(rng.integers(0, n)), int(rng.integers(0, n))
            l = int(min(rng.integers(4, 301), n - s, n - d))
            a[d:d + l] = a[s:s + l].copy()
    return a.tobytes()


def long_range(n, seed=6, base_frac=0.5, mutate_every=0):
    """First base_frac of the buffer is text-like; the rest repeats it (optionally mutated)."""
    nb = max(1, int(n * base_frac))
    base = np.frombuffer(text_like(nb, seed), dtype=np.uint8)
    reps = -(-n // nb)
    a = np.tile(base, reps)[:n].copy()
    if mutate_every:
        rng = np.random.default_rng(seed + 100)
        pos = np.arange(nb + mutate_every // 2, n, mutate_every)
        a[pos] = rng.integers(0, 256, size=len(pos), dtype=np.uint8)
    return a.tobytes()


KINDS = {
    "text": text_like, "random": random_bytes, "few": few_symbols, "phrases": phrase_mix,
    "sparse": sparse_repeats, "zeros": lambda n, seed=0: bytes(n), "longrange": long_range,
}


def text_alnum(n, seed=1, nwords=5000):
    """Word-list pseudo text over the 62-symbol alphabet bench.py uses (the tag space does not collapse)."""
    abc = np.frombuffer(b"abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ0123456789", dtype=np.uint8)
    rng = np.random.default_rng(seed)
    lens = rng.integers(2, 10, size=nwords)
    words = [abc[rng.integers(0, 62, size=int(l))].tobytes() + b" " for l in lens]
    out = bytearray()
    while len(out) < n:
        idx = rng.integers(0, nwords, size=65536)
        out += b"".join(words[i] for i in idx)
    return bytes(out[:n])


def cfg3(n, base, seed=1, mutate_every=65536):
    """BASELINE config 3 shape: a seeded text base block repeated to n bytes, one seeded byte
    mutation per 64 KiB in every copy after the first."""
    b = np.frombuffer(text_alnum(base, seed), dtype=np.uint8)
    a = np.tile(b, -(-n // base))[:n].copy()
    rng = np.random.default_rng(seed + 7919)
    cells = (n - base) // mutate_every
    if cells > 0:
        pos = base + np.arange(cells, dtype=np.int64) * mutate_every + rng.integers(0, mutate_every, size=cells)
        a[pos] = rng.integers(0, 256, size=cells, dtype=np.uint8)
    return a.tobytes()


def victim_mover(n, hash_index, seed=1, every=65536, count_limit=None):
    """Random bytes with, every `every` bytes, a fresh PERMUTATION of one fixed 31-byte multiset.  rzip's tag is
    the XOR of per-byte table values, blind to order: all those windows carry the same tag but never match
    (31 equal bytes are needed), so they pile up in the hash table until max_chain_len of them are met and
    insert_hash()'s round-robin victim counter (the static of src/rzip.c:308) advances.  The multiset is chosen
    so that the tag has its 20 low bits set: it stays a candidate however far the tag masks have tightened."""
    rng = np.random.default_rng(seed)
    hx = np.array(hash_index, dtype=np.uint64)
    while True:
        cand = rng.integers(0, 256, size=(1 << 18, 31), dtype=np.uint8)
        t = np.bitwise_xor.reduce(hx[cand], axis=1)
        hit = np.nonzero((t & np.uint64(0xFFFFF)) == np.uint64(0xFFFFF))[0]
        if len(hit):
            ms = cand[hit[0]]
            break
    a = rng.integers(0, 256, size=n, dtype=np.uint8)
    k = 0
    for at in range(every // 2, n - 64, every):
        if count_limit is not None and k >= count_limit:
            break
        a[at:at + 31] = rng.permutation(ms)
        k += 1
    return a.tobytes()


def source_tree_tar(copies, tree_bytes, seed=1, variants=1):
    """BASELINE config 4 shape: an (uncompressed) tar of `copies` copies of one seeded synthetic source tree --
    a few hundred text files of 1..64 KiB (identifiers, punctuation, indentation, newlines) under per-copy
    directory names, so headers differ and contents repeat at tree distance.  variants > 1 (the full-size run: the
    line-by-line generator is slow): only tree_bytes / variants are generated, the rest of the tree are copies of
    those files under seeded permutations of the alphabet -- different text of the same make."""
    import io
    import tarfile
    rng = np.random.default_rng(seed)
    words = [w for w in text_alnum(40000, seed + 1).split(b" ") if w]
    punct = [b"(", b")", b";", b" = ", b", ", b"{", b"}", b"->", b"[i]", b" + ", b"// "]
    files = []
    total = 0
    tree_bytes_wanted = tree_bytes
    tree_bytes = tree_bytes // max(1, variants)
    while total < tree_bytes:
        size = int(rng.integers(1024, 65536))
        lines, got = [], 0
        while got < size:
            k = int(rng.integers(2, 9))
            idx = rng.integers(0, len(words), size=k)
            pidx = rng.integers(0, len(punct), size=k)
            line = b"\t" * int(rng.integers(0, 4)) + b"".join(words[i] + punct[p] for i, p in zip(idx, pidx)) + b"\n"
            lines.append(line)
            got += len(line)
        body = b"".join(lines)[:size]
        files.append(("src/mod%03d/file%04d.c" % (len(files) // 16, len(files)), body))
        total += size
    if variants > 1:
        base = list(files)
        letters = np.frombuffer(b"abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ0123456789", dtype=np.uint8)
        for v in range(1, variants):
            table = np.arange(256, dtype=np.uint8)
            table[letters] = rng.permutation(letters)
            tb = table.tobytes()
            for name, body in base:
                if total >= tree_bytes_wanted:
                    break
                files.append(("src/mod%03d/file%04d.c" % (len(files) // 16, len(files)), body.translate(tb)))
                total += len(body)
    out = io.BytesIO()
    with tarfile.open(fileobj=out, mode="w", format=tarfile.USTAR_FORMAT) as tf:
        for c in range(copies):
            for name, body in files:
                ti = tarfile.TarInfo("tree_%02d/%s" % (c, name))
                ti.size = len(body)
                ti.mtime = 1700000000 + c
                tf.addfile(ti, io.BytesIO(body))
    return out.getvalue()
