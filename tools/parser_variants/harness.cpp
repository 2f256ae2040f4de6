// Host parser timing harness (no GPU, no Python): reads a block, its list counts and its packed lists from files,
// runs T concurrent encoders on it (each its own output buffer, like the pipeline's encoder threads), prints wall
// time, per-thread time and -- where the kernel lets this process open them -- hardware counters per input byte
// (cycles, instructions, branch misses, L1D / LLC misses) summed over the threads.
// usage: harness <dir with d.bin c.bin p2.bin> <threads> [reps]
// PV_SAMPLE=cycles|instructions|branch-misses (one thread only): the kernel samples the instruction pointer every
// PV_PERIOD events (perf_event_open + its mmap ring, what `perf record` does; there is no perf in the image) and the
// histogram goes to PV_SAMPLE_OUT (address count per line) for tools/parser_variants/lines.py.
#include <linux/perf_event.h>
#include <sys/ioctl.h>
#include <sys/mman.h>
#include <map>
#include <sys/syscall.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "lzma_enc.h"
using namespace lrzgpu;

static std::vector<uint8_t> rd(const std::string &p)
{
	FILE *f = fopen(p.c_str(), "rb");
	if (!f) {
		perror(p.c_str());
		exit(2);
	}
	fseek(f, 0, SEEK_END);
	long n = ftell(f);
	fseek(f, 0, SEEK_SET);
	std::vector<uint8_t> v(n);
	if (fread(v.data(), 1, n, f) != (size_t)n)
		abort();
	fclose(f);
	return v;
}

struct Counter {
	const char *name;
	uint32_t type;
	uint64_t config;
};
static const Counter kCounters[] = {
	{"cycles", PERF_TYPE_HARDWARE, PERF_COUNT_HW_CPU_CYCLES},
	{"instructions", PERF_TYPE_HARDWARE, PERF_COUNT_HW_INSTRUCTIONS},
	{"branches", PERF_TYPE_HARDWARE, PERF_COUNT_HW_BRANCH_INSTRUCTIONS},
	{"branch-misses", PERF_TYPE_HARDWARE, PERF_COUNT_HW_BRANCH_MISSES},
	{"L1D-read-misses", PERF_TYPE_HW_CACHE, PERF_COUNT_HW_CACHE_L1D | (PERF_COUNT_HW_CACHE_OP_READ << 8) | (PERF_COUNT_HW_CACHE_RESULT_MISS << 16)},
	{"cache-misses", PERF_TYPE_HARDWARE, PERF_COUNT_HW_CACHE_MISSES},
};
constexpr int kNC = sizeof(kCounters) / sizeof(kCounters[0]);

static int open_counter(const Counter &c)
{
	perf_event_attr a;
	memset(&a, 0, sizeof a);
	a.type = c.type;
	a.size = sizeof a;
	a.config = c.config;
	a.exclude_kernel = 1;
	a.exclude_hv = 1;
	a.disabled = 1;
	return (int)syscall(SYS_perf_event_open, &a, 0, -1, -1, 0); // this thread, any CPU
}

// ---- sampling: IP every `period` events of one hardware counter, this thread ----------------------------------------
struct Sampler {
	bool ibs_raw = false; // the IBS registers of every sample are kept too (latencies, cache misses, mispredicts)
	int fd = -1;
	void *ring = nullptr;
	size_t ring_bytes = 0;
	bool start(const char *what, uint64_t period)
	{
		perf_event_attr a;
		memset(&a, 0, sizeof a);
		a.type = PERF_TYPE_HARDWARE;
		a.size = sizeof a;
		a.config = !strcmp(what, "instructions") ? PERF_COUNT_HW_INSTRUCTIONS : !strcmp(what, "branch-misses") ? PERF_COUNT_HW_BRANCH_MISSES :
			   !strcmp(what, "cache-misses") ? PERF_COUNT_HW_CACHE_MISSES : PERF_COUNT_HW_CPU_CYCLES;
		if (!strcmp(what, "l1d-misses")) {
			a.type = PERF_TYPE_HW_CACHE;
			a.config = PERF_COUNT_HW_CACHE_L1D | (PERF_COUNT_HW_CACHE_OP_READ << 8) | (PERF_COUNT_HW_CACHE_RESULT_MISS << 16);
		}
		bool ibs = false;
		if (!strcmp(what, "ibs-op") || !strcmp(what, "ibs-cycles")) {
			// AMD instruction based sampling: the sampled op's own address, no skid.  ibs-op counts dispatched ops
			// (an instruction-weighted profile), ibs-cycles counts cycles.
			FILE *t = fopen("/sys/bus/event_source/devices/ibs_op/type", "r");
			int ty = -1;
			if (!t || fscanf(t, "%d", &ty) != 1 || ty < 0) {
				fprintf(stderr, "no ibs_op PMU here\n");
				return false;
			}
			fclose(t);
			a.type = (uint32_t)ty;
			a.config = !strcmp(what, "ibs-op") ? (1ull << 19) : 0;
			a.sample_period = (period + 15) & ~15ull;
			ibs = true;
			ibs_raw = true;
		}
		if (!strncmp(what, "raw:", 4)) { // raw:<hex event code of this CPU's PMU>
			a.type = PERF_TYPE_RAW;
			a.config = strtoull(what + 4, nullptr, 16);
		}
		if (!a.sample_period)
			a.sample_period = period;
		a.sample_type = PERF_SAMPLE_IP | (ibs_raw ? PERF_SAMPLE_RAW : 0);
		a.exclude_kernel = ibs ? 0 : 1; // (the IBS PMU takes no privilege filter)
		a.exclude_hv = ibs ? 0 : 1;
		a.disabled = 1;
		a.precise_ip = 0;
		fd = (int)syscall(SYS_perf_event_open, &a, 0, -1, -1, 0);
		if (fd < 0) {
			perror("perf_event_open (sampling)");
			return false;
		}
		ring_bytes = (size_t)(1 + 4096) * 4096; // 16 MiB of samples
		ring = mmap(nullptr, ring_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
		if (ring == MAP_FAILED) {
			ring_bytes = (size_t)(1 + 128) * 4096;
			ring = mmap(nullptr, ring_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
		}
		if (ring == MAP_FAILED) {
			close(fd);
			fd = -1;
			return false;
		}
		ioctl(fd, PERF_EVENT_IOC_ENABLE, 0);
		return true;
	}
	void stop_and_dump(const char *path)
	{
		ioctl(fd, PERF_EVENT_IOC_DISABLE, 0);
		auto *meta = (perf_event_mmap_page *)ring;
		const uint64_t head = meta->data_head, size = ring_bytes - 4096;
		const uint8_t *base = (const uint8_t *)ring + 4096;
		// per address: samples, and from the IBS registers of each sampled op (AMD PPR, IbsOpData / IbsOpData3): cycles
		// from tagging to retirement, from completion to retirement (waiting for OLDER ops), loads that missed L1 and
		// their miss latency, mispredicted branches
		struct Acc {
			uint64_t n = 0, tag_ret = 0, comp_ret = 0, dc_miss = 0, dc_lat = 0, br_misp = 0;
		};
		std::map<uint64_t, Acc> hist;
		uint64_t pos = head > size ? head - size : 0, n = 0; // (an overrun keeps the newest)
		auto rd = [&](uint64_t at, void *dst, size_t len) {
			for (size_t k = 0; k < len; k++)
				((uint8_t *)dst)[k] = base[(at + k) % size];
		};
		while (pos + sizeof(perf_event_header) <= head) {
			perf_event_header h;
			rd(pos, &h, sizeof h);
			if (!h.size)
				break;
			if (h.type == PERF_RECORD_SAMPLE && h.size >= sizeof h + 8) {
				uint64_t ip = 0;
				rd(pos + sizeof h, &ip, 8);
				Acc &a = hist[ip];
				a.n++;
				n++;
				if (ibs_raw && h.size >= sizeof h + 8 + 4 + 4 + 5 * 8) {
					uint32_t rsz = 0;
					rd(pos + sizeof h + 8, &rsz, 4);
					uint64_t regs[5]; // IbsOpCtl, IbsOpRip, IbsOpData, IbsOpData2, IbsOpData3 (after the 4-byte capability word)
					if (rsz >= 4 + sizeof regs) {
						rd(pos + sizeof h + 8 + 4 + 4, regs, sizeof regs);
						const uint64_t d1 = regs[2], d3 = regs[4];
						a.comp_ret += d1 & 0xFFFF;
						a.tag_ret += (d1 >> 16) & 0xFFFF;
						a.br_misp += (d1 >> 36) & 1;
						if ((d3 & 1) && ((d3 >> 7) & 1)) { // a load that missed the data cache
							a.dc_miss++;
							a.dc_lat += (d3 >> 32) & 0xFFFF;
						}
					}
				}
			}
			pos += h.size;
		}
		FILE *f = fopen(path, "w");
		for (auto &kv : hist) {
			const Acc &a = kv.second;
			if (ibs_raw)
				fprintf(f, "%llx %llu %llu %llu %llu %llu %llu\n", (unsigned long long)kv.first, (unsigned long long)a.n, (unsigned long long)a.tag_ret,
					(unsigned long long)a.comp_ret, (unsigned long long)a.dc_miss, (unsigned long long)a.dc_lat, (unsigned long long)a.br_misp);
			else
				fprintf(f, "%llx %llu\n", (unsigned long long)kv.first, (unsigned long long)a.n);
		}
		fclose(f);
		fprintf(stderr, "%llu samples -> %s%s\n", (unsigned long long)n, path, head > size ? " (ring overran: raise PV_PERIOD)" : "");
	}
};

int main(int argc, char **argv)
{
	if (argc < 3) {
		fprintf(stderr, "usage: %s <dir> <threads> [reps]\n", argv[0]);
		return 2;
	}
	const std::string dir = argv[1];
	const int T = atoi(argv[2]), reps = argc > 3 ? atoi(argv[3]) : 2;
	const auto d = rd(dir + "/d.bin"), c = rd(dir + "/c.bin"), p = rd(dir + "/p2.bin");
	LzmaParams prm;
	prm.level = 7;
	prm.dict_size = 1u << 25;
	prm.fb = 64;
	MatchLists ml;
	ml.counts = c.data();
	ml.pairs = (const uint32_t *)p.data();
	ml.tail_flags = ml.packed = true;
	double best_wall = 1e9, best_cpu = 1e9;
	unsigned long long hash = 0;
	size_t out_len = 0;
	std::vector<double> ctr_best(kNC, 0);
	bool have_ctr = false;
	for (int r = 0; r < reps; r++) {
		std::vector<double> per(T);
		std::vector<std::vector<uint64_t>> ctr(T, std::vector<uint64_t>(kNC, 0));
		std::vector<std::thread> th;
		const auto t0 = std::chrono::steady_clock::now();
		for (int i = 0; i < T; i++)
			th.emplace_back([&, i] {
				std::vector<uint8_t> out(d.size() + d.size() / 3 + 4096);
				int fd[kNC];
				for (int k = 0; k < kNC; k++) {
					fd[k] = getenv("PV_SAMPLE") ? -1 : open_counter(kCounters[k]);
					if (fd[k] >= 0)
						ioctl(fd[k], PERF_EVENT_IOC_ENABLE, 0);
				}
				size_t ol = 0;
				Sampler smp;
				const char *what = getenv("PV_SAMPLE");
				const bool sampling = what && T == 1 && r == reps - 1 && smp.start(what, getenv("PV_PERIOD") ? strtoull(getenv("PV_PERIOD"), nullptr, 0) : 200003);
				const auto a = std::chrono::steady_clock::now();
				const int rc = lzma_encode_block(prm, d.data(), d.size(), ml, out.data(), out.size(), &ol);
				per[i] = std::chrono::duration<double>(std::chrono::steady_clock::now() - a).count();
				if (sampling)
					smp.stop_and_dump(getenv("PV_SAMPLE_OUT") ? getenv("PV_SAMPLE_OUT") : "/tmp/pv_samples.txt");
				for (int k = 0; k < kNC; k++)
					if (fd[k] >= 0) {
						ioctl(fd[k], PERF_EVENT_IOC_DISABLE, 0);
						if (read(fd[k], &ctr[i][k], 8) != 8)
							ctr[i][k] = 0;
						close(fd[k]);
					}
				if (rc != 0)
					abort();
				if (i == 0) {
					unsigned long long h = 1469598103934665603ull;
					for (size_t j = 0; j < ol; j++)
						h = (h ^ out[j]) * 1099511628211ull;
					hash = h;
					out_len = ol;
				}
			});
		for (auto &t : th)
			t.join();
		const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
		double sum = 0;
		for (double x : per)
			sum += x;
		if (sum / T < best_cpu) {
			best_cpu = sum / T;
			best_wall = wall;
			for (int k = 0; k < kNC; k++) {
				double s = 0;
				for (int i = 0; i < T; i++)
					s += (double)ctr[i][k];
				ctr_best[k] = s / T;
				if (s > 0)
					have_ctr = true;
			}
		}
	}
	const double mib = d.size() / 1048576.0;
	printf("T=%2d  %.3f s per thread  %.2f MiB/s per thread  %.1f MiB/s total  wall %.3f s  (out %zu, hash %016llx)\n", T, best_cpu, mib / best_cpu,
	       T * mib / best_wall, best_wall, out_len, hash);
	if (have_ctr) {
		printf("      per input byte:");
		for (int k = 0; k < kNC; k++)
			printf("  %s %.2f", kCounters[k].name, ctr_best[k] / d.size());
		printf("\n");
	} else
		printf("      (no hardware counters: perf_event_open refused)\n");
	return 0;
}
