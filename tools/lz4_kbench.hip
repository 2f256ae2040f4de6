// lz4_kbench.hip -- stand-alone timing of the lz4 gate kernel on synthetic text (developer tool).
// usage: lz4_kbench [block MiB] [blocks]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../lrzip-next_amd/csrc/lz4_gate.h"

using namespace lrzgpu;

int main(int argc, char **argv)
{
	const size_t mib = argc > 1 ? atoi(argv[1]) : 16;
	const int blocks = argc > 2 ? atoi(argv[2]) : 8;
	const size_t n = mib << 20;
	// word-salad text: 5000 random words over [a-zA-Z0-9]
	std::vector<std::vector<uint8_t>> words(5000);
	uint64_t s = 88172645463325252ULL;
	auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
	const char *alpha = "abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ0123456789";
	for (auto &w : words) {
		int len = 2 + rnd() % 9;
		for (int i = 0; i < len; i++)
			w.push_back(alpha[rnd() % 62]);
	}
	std::vector<uint8_t> h(n * blocks + 64);
	size_t p = 0;
	while (p < n * blocks) {
		auto &w = words[rnd() % 5000];
		for (uint8_t c : w)
			if (p < n * blocks)
				h[p++] = c;
		if (p < n * blocks)
			h[p++] = ' ';
	}
	uint8_t *d;
	hipMalloc(&d, h.size());
	hipMemcpy(d, h.data(), h.size(), hipMemcpyHostToDevice);
	std::vector<Lz4Job> jobs(blocks);
	for (int b = 0; b < blocks; b++)
		jobs[b] = {d + b * n, (int)n, (int)n + 1};
	Lz4Job *dj;
	int *dr;
	hipMalloc(&dj, blocks * sizeof(Lz4Job));
	hipMalloc(&dr, blocks * sizeof(int));
	hipMemcpy(dj, jobs.data(), blocks * sizeof(Lz4Job), hipMemcpyHostToDevice);
	hipEvent_t e0, e1;
	hipEventCreate(&e0);
	hipEventCreate(&e1);
	for (int rep = 0; rep < 2; rep++) {
		hipEventRecord(e0, 0);
		if (lz4_sizes_device(dj, blocks, dr, 0) != 0) {
			fprintf(stderr, "launch failed\n");
			return 1;
		}
		hipEventRecord(e1, 0);
		hipEventSynchronize(e1);
		float ms;
		hipEventElapsedTime(&ms, e0, e1);
		std::vector<int> r(blocks);
		hipMemcpy(r.data(), dr, blocks * sizeof(int), hipMemcpyDeviceToHost);
		printf("%d blocks x %zu MiB: %.1f ms (%.2f MB/s per wave), sizes %d %d\n", blocks, mib, ms,
		       (double)n / 1048576.0 / (ms / 1e3), r[0], r[blocks - 1]);
	}
	return 0;
}
