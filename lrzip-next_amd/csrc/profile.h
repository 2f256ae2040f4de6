// profile.h -- per-kernel timing (HIP events on the launching stream) accumulated for bench.py.
#pragma once
#include <hip/hip_runtime.h>

#include <mutex>
#include <utility>
#include <vector>

#include "../../include/lrzgpu.h"

namespace lrzgpu {

enum ProfileKind { PK_TAG_SCAN = 0, PK_RESOLVE, PK_CRC, PK_GATHER, PK_LZ4, PK_MF_BT, PK_MF_TOTAL, PK_LONG_COMPARE, PK_COUNT };

struct ProfileStore {
	std::mutex mu;
	lrzgpu_profile p{};
	// when every launch of a kind started and ended, in ms since `base` (recorded by lrzgpu_profile_reset): launches of
	// one kind run side by side on different streams (a resolver per chunk, a finder per GPU slot), so the SUM of their
	// durations says nothing about the wall time they cover -- the union of these intervals does
	// Bounded: a process that reset once and then compresses for days keeps the first kMaxIntervals launches of a kind
	// (2 MB per kind) and counts the rest (`dropped`; union_ms / peak_concurrency then describe the recorded ones).
	static constexpr size_t kMaxIntervals = (size_t)1 << 18;
	hipEvent_t base = nullptr;
	std::vector<std::pair<float, float>> iv[PK_COUNT];
	long long dropped[PK_COUNT] = {0};
	static ProfileStore &get()
	{
		static ProfileStore s;
		return s;
	}
};

// Scoped pair of events around work submitted to `s`; call done() after the stream was synchronised.
struct EventTimer {
	hipEvent_t a = nullptr, b = nullptr;
	hipStream_t s;
	explicit EventTimer(hipStream_t st) : s(st)
	{
		if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess)
			a = b = nullptr;
		if (a)
			(void)hipEventRecord(a, s);
	}
	void stop()
	{
		if (b)
			(void)hipEventRecord(b, s);
	}
	double ms() // after synchronisation
	{
		float t = 0;
		if (a && b && hipEventElapsedTime(&t, a, b) != hipSuccess)
			t = 0;
		return t;
	}
	// duration, and the launch noted as an interval of `kind` (ProfileStore::mu must be held by the caller)
	double ms_noted(ProfileStore &ps, int kind)
	{
		const double d = ms();
		float at = 0;
		if (ps.base && a && ps.iv[kind].size() >= ProfileStore::kMaxIntervals)
			ps.dropped[kind]++;
		else if (ps.base && a && hipEventElapsedTime(&at, ps.base, a) == hipSuccess)
			ps.iv[kind].push_back(std::make_pair(at, at + (float)d));
		else
			(void)hipGetLastError();
		return d;
	}
	~EventTimer()
	{
		if (a)
			(void)hipEventDestroy(a);
		if (b)
			(void)hipEventDestroy(b);
	}
};

} // namespace lrzgpu
