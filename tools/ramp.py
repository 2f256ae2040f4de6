#!/usr/bin/env python3
"""The first seconds of one traced run (LRZGPU_TRACE=2 stderr): what the encoders are doing while the chunks' first
blocks are still being scanned -- per quarter second: encoders inside a block, of those waiting for the next part of it;
and the story of chunk 0's first block (finder runs on its prefixes, what the encoder was handed when)."""
import sys
rows = []
for l in open(sys.argv[1]):
    if l.startswith("ev "):
        f = l.split()
        rows.append((float(f[1]), f[2], int(f[4]), int(f[6]), int(f[8]), int(f[10]), float(f[12]) if len(f) > 12 else 0.0))
T = float(sys.argv[2]) if len(sys.argv) > 2 else 6.0
enc = {}
for t, what, c, s, off, ln, w in rows:
    k = (c, s, off)
    if what == "enc_start":
        enc[k] = [t, None, []]
    elif what == "enc_end" and k in enc:
        enc[k][1] = t
    elif what == "rest" and k in enc:
        enc[k][2].append((t - w, t))
print("t      encoders inside a block   of those waiting for more of it")
t = 0.0
while t < T:
    inside = sum(1 for a, b, _ in enc.values() if a <= t < (b if b is not None else 1e9))
    waiting = sum(1 for a, b, ws in enc.values() for (x, y) in ws if x <= t < y)
    print("%5.2f  %3d  %3d" % (t, inside, waiting))
    t += 0.25
print("chunk 0, first block:")
for t, what, c, s, off, ln, w in rows:
    if c == 0 and s == 1 and off == 0 and t < T:
        print("  %6.3f %-12s len %10d%s" % (t, what, ln, "  (waited %.3f)" % w if what == "rest" else ""))
