"""Seeded generator of the SURVEY.md Appendix-A probe input ``syn64.bin``.

The survey recorded reference outputs (sha256 of the .lrz files the reference
binary produced) for exactly this input; reproducing the input lets the oracle
be pinned against those recorded reference answers without the reference.
"""
import hashlib
import random
import sys


def syn_text(n_bytes: int, seed: int = 1) -> bytes:
    """``n_bytes`` of pseudo-text: 5000 random lowercase words, space separated."""
    random.seed(seed)
    letters = bytes(range(ord("a"), ord("z") + 1))
    words = [bytes(random.choices(letters, k=random.randint(2, 9))) for _ in range(5000)]
    out = bytearray()
    while len(out) < n_bytes:
        out += random.choice(words) + b" "
    return bytes(out[:n_bytes])


def syn64() -> bytes:
    half = syn_text(32 << 20, 1)
    return half + half


if __name__ == "__main__":
    data = syn64()
    print(hashlib.sha256(data).hexdigest(), len(data))
    if len(sys.argv) > 1:
        open(sys.argv[1], "wb").write(data)
