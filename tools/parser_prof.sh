#!/bin/bash
# Cycle laps of the host LZMA parser on the data of tools/parser_bench.py (run that first: it caches the lists).
set -e
cd "$(dirname "$0")/.."
mkdir -p /tmp/prof
python3 - <<'PY'
import pickle, glob
f = sorted(glob.glob('/tmp/parser_bench_alnum_*_7.pkl'))[-1]
data, counts, pairs, want = pickle.load(open(f, 'rb'))
open('/tmp/prof/d.bin', 'wb').write(data); counts.tofile('/tmp/prof/c.bin'); pairs.tofile('/tmp/prof/p0.bin')
import sys, os
sys.path.insert(0, 'tests')
from conftest import load_bindings
B = load_bindings()
for fmt in (1, 2):
    B.format_lists(data, counts, pairs, fmt).tofile('/tmp/prof/p%d.bin' % fmt)
PY
cat > /tmp/prof/main.cpp <<'CPP'
#include "lzma_enc.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <chrono>
using namespace lrzgpu;
int main(int argc, char **argv)
{
	auto rd = [&](const char *p) { FILE *f = fopen(p, "rb"); fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET); std::vector<uint8_t> v(n); if (fread(v.data(), 1, n, f) != (size_t)n) abort(); fclose(f); return v; };
	const int fmt = argc > 1 ? atoi(argv[1]) : 0;
	char pn[64]; snprintf(pn, sizeof pn, "/tmp/prof/p%d.bin", fmt);
	auto d = rd("/tmp/prof/d.bin"), c = rd("/tmp/prof/c.bin"), p = rd(pn);
	LzmaParams prm; prm.level = 7; prm.dict_size = 1u << 25; prm.fb = 64;
	MatchLists ml; ml.counts = c.data(); ml.pairs = (const uint32_t *)p.data(); ml.tail_flags = fmt != 0; ml.packed = fmt == 2;
	std::vector<uint8_t> out(d.size() + d.size() / 3 + 4096); size_t ol = 0;
	auto t0 = std::chrono::steady_clock::now();
	int r = 0; double best = 1e9;
	for (int k = 0; k < (argc > 2 ? atoi(argv[2]) : 3); k++) {
		t0 = std::chrono::steady_clock::now();
		r = lzma_encode_block(prm, d.data(), d.size(), ml, out.data(), out.size(), &ol);
		double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
		if (s < best) best = s;
	}
	printf("fmt %d rc %d out %zu  best %.3f s  %.2f MiB/s\n", fmt, r, ol, best, d.size() / 1048576.0 / best);
	return 0;
}
CPP
g++ -O3 -march=x86-64-v4 -std=c++17 ${PROF:+-DLZMA_PARSER_PROF} -Ilrzip-next_amd/csrc -c -o /tmp/prof/p4.o lrzip-next_amd/csrc/lzma_parser.cpp
g++ -O3 -march=x86-64-v3 -std=c++17 ${PROF:+-DLZMA_PARSER_PROF} -Ilrzip-next_amd/csrc -o /tmp/prof/prof /tmp/prof/main.cpp lrzip-next_amd/csrc/lzma_parser.cpp /tmp/prof/p4.o
for f in ${FORMATS:-0 2}; do /tmp/prof/prof $f; done
