#!/usr/bin/env python3
"""lines.py <binary> <samples file> [min %]: the sampled instruction pointers of tools/parser_variants/harness.cpp per
source line (addr2line; the innermost inlined frame) and per function."""
import collections, subprocess, sys
binary, samples = sys.argv[1], sys.argv[2]
floor = float(sys.argv[3]) if len(sys.argv) > 3 else 0.4
rows = [l.split()[:2] for l in open(samples)]
addrs = [r[0] for r in rows]
cnt = [int(r[1]) for r in rows]
out = subprocess.run(["addr2line", "-e", binary, "-f", "-C"] + addrs, capture_output=True, text=True).stdout.splitlines()
lines, funcs, tot = collections.Counter(), collections.Counter(), sum(cnt)
for i, c in enumerate(cnt):
    lines[out[2 * i + 1].split("/")[-1].split(" ")[0]] += c
    funcs[out[2 * i][:100]] += c
print("samples %d" % tot)
for k, v in funcs.most_common(12):
    print("%6.2f%% %s" % (100.0 * v / tot, k))
print()
def key(kv):
    f, _, ln = kv[0].partition(":")
    return (f, int(ln) if ln.isdigit() else 0)
for k, v in sorted(lines.items(), key=key):
    if 100.0 * v / tot >= floor:
        print("%6.2f%% %s" % (100.0 * v / tot, k))
