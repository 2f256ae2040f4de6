#include "lzma_enc.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <chrono>
#include <sys/mman.h>
#include <cstring>
using namespace lrzgpu;
void sampler_start(); void sampler_dump(const char*);
int main(int argc, char **argv)
{
	auto rd = [&](const char *p) { FILE *f = fopen(p, "rb"); fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET); std::vector<uint8_t> v(n); if (fread(v.data(), 1, n, f) != (size_t)n) abort(); fclose(f); return v; };
	const int fmt = argc > 1 ? atoi(argv[1]) : 0;
	char pn[64]; snprintf(pn, sizeof pn, "/tmp/prof/p%d.bin", fmt);
	auto d = rd("/tmp/prof/d.bin"), c = rd("/tmp/prof/c.bin"), p = rd(pn);
	uint8_t *dd = d.data();
	if (getenv("HUGE")) { size_t sz = (d.size() + (2u<<20) - 1) & ~(size_t)((2u<<20)-1); dd = (uint8_t*)aligned_alloc(2u<<20, sz); madvise(dd, sz, MADV_HUGEPAGE); memcpy(dd, d.data(), d.size()); }
	LzmaParams prm; prm.level = 7; prm.dict_size = 1u << 25; prm.fb = 64;
	MatchLists ml; ml.counts = c.data(); ml.pairs = (const uint32_t *)p.data(); ml.tail_flags = fmt != 0; ml.packed = fmt == 2;
	std::vector<uint8_t> out(d.size() + d.size() / 3 + 4096); size_t ol = 0;
	int r = 0; double best = 1e9;
	if (getenv("SAMPLE")) sampler_start();
	for (int k = 0; k < (argc > 2 ? atoi(argv[2]) : 3); k++) {
		auto t0 = std::chrono::steady_clock::now();
		r = lzma_encode_block(prm, dd, d.size(), ml, out.data(), out.size(), &ol);
		double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
		if (s < best) best = s;
	}
	if (getenv("SAMPLE")) sampler_dump("/tmp/prof/samples.txt");
	unsigned long long h = 1469598103934665603ull; for (size_t i = 0; i < ol; i++) h = (h ^ out[i]) * 1099511628211ull;
	printf("fmt %d rc %d out %zu hash %016llx best %.3f s  %.2f MiB/s\n", fmt, r, ol, h, best, d.size() / 1048576.0 / best);
	return 0;
}
