#!/bin/bash
# Where the host parser's cycles go, line by line, on THIS host (no perf in the image: a SIGPROF sampler of the instruction
# pointer + addr2line).  A 64 MiB block of the bench text, lists from the GPU finder in the pipeline's packed format;
# the parser is built without the recorded profile here (-g, frame info) -- the shape of the profile is what is wanted.
# Output: gpurun_out/parser_sample/lines.txt (per source line) and funcs.txt.
set -e
cd "$(dirname "$0")/.."
OUT=gpurun_out/parser_sample; mkdir -p $OUT /tmp/prof
python3 - <<'PY'
import sys, os
sys.path.insert(0, 'tests')
import datagen
from conftest import load_bindings
B = load_bindings()
n = 64 << 20
data = datagen.text_alnum(n, seed=1)
counts, pairs = B.lzma_match_lists(data, dict_size=1 << 25, fb=64, cut=48, per_pos=12)
open('/tmp/prof/d.bin', 'wb').write(data); counts.tofile('/tmp/prof/c.bin')
B.format_lists(data, counts, pairs, 2).tofile('/tmp/prof/p2.bin')
PY
g++ -O3 -g -march=x86-64-v4 -std=c++17 -fno-pie -Ilrzip-next_amd/csrc -c -o /tmp/prof/p4.o lrzip-next_amd/csrc/lzma_parser.cpp
g++ -O3 -g -march=x86-64-v3 -std=c++17 -fno-pie -no-pie -Ilrzip-next_amd/csrc -o /tmp/prof/prof2 tools/parser_sampler/main.cpp tools/parser_sampler/sampler.cpp lrzip-next_amd/csrc/lzma_parser.cpp /tmp/prof/p4.o
/tmp/prof/prof2 2 1 | tee $OUT/speed.txt
SAMPLE=1 /tmp/prof/prof2 2 2 | tee -a $OUT/speed.txt
python3 - <<'PY'
import subprocess, collections
rows = [l.split() for l in open('/tmp/prof/samples.txt')]
addrs = [r[0] for r in rows]; cnt = [int(r[1]) for r in rows]
out = subprocess.run(['addr2line', '-e', '/tmp/prof/prof2', '-f', '-C', '-i'] + addrs, capture_output=True, text=True).stdout.splitlines()
# -i prints the inline chain: innermost first; we want, per address, the innermost frame (line) and the outermost function
plain = subprocess.run(['addr2line', '-e', '/tmp/prof/prof2', '-f', '-C'] + addrs, capture_output=True, text=True).stdout.splitlines()
lines = collections.Counter(); funcs = collections.Counter(); tot = sum(cnt)
for i, c in enumerate(cnt):
    lines[plain[2 * i + 1].split('/')[-1]] += c
    funcs[plain[2 * i][:90]] += c
with open('gpurun_out/parser_sample/lines.txt', 'w') as f:
    f.write("samples %d\n" % tot)
    for k, v in lines.most_common(120):
        f.write("%5.2f%% %s\n" % (100.0 * v / tot, k))
with open('gpurun_out/parser_sample/funcs.txt', 'w') as f:
    for k, v in funcs.most_common(40):
        f.write("%5.2f%% %s\n" % (100.0 * v / tot, k))
PY
head -45 $OUT/lines.txt
