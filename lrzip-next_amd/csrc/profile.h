// profile.h -- per-kernel timing (HIP events on the launching stream) accumulated for bench.py.
#pragma once
#include <hip/hip_runtime.h>

#include <mutex>

#include "../../include/lrzgpu.h"

namespace lrzgpu {

struct ProfileStore {
	std::mutex mu;
	lrzgpu_profile p{};
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
	~EventTimer()
	{
		if (a)
			(void)hipEventDestroy(a);
		if (b)
			(void)hipEventDestroy(b);
	}
};

} // namespace lrzgpu
