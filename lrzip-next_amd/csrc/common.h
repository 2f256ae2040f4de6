// common.h -- small helpers shared by the host-side translation units.
#pragma once
#include <time.h>
#include <hip/hip_runtime.h>

#include <cstdint>

#include "lzma_enc.h"

namespace lrzgpu {
// Wait without burning a host core.  On this runtime (ROCm 7.2) hipStreamSynchronize(), hipEventSynchronize() -- WITH
// the hipEventBlockingSync flag too -- and a hipMemcpyAsync() into pageable memory all busy-wait unless the PROCESS set
// hipDeviceScheduleBlockingSync (tools/waitprobe: 300 ms of thread CPU for a 300 ms kernel, every variant), and a
// library has no business flipping a process-wide device flag.  The host cores are what the LZMA encoders need (16 CPUs
// of quota, ~94 % used), so: query + sleep.  40 us of queries catch the short kernels, then the sleeps grow from
// 30 us to 400 us -- the wake-up is at most that late, against launches of milliseconds to seconds.
inline hipError_t event_wait(hipEvent_t ev)
{
	hipError_t q = hipEventQuery(ev);
	if (q != hipErrorNotReady)
		return q;
	timespec t0, t;
	clock_gettime(CLOCK_MONOTONIC, &t0);
	do { // ~40 us of queries: the short launches (a gate batch, a tile CRC) end inside it
		q = hipEventQuery(ev);
		clock_gettime(CLOCK_MONOTONIC, &t);
	} while (q == hipErrorNotReady && (t.tv_sec - t0.tv_sec) * 1000000000L + (t.tv_nsec - t0.tv_nsec) < 40000);
	long ns = 30000;
	while (q == hipErrorNotReady) {
		timespec ts{0, ns};
		nanosleep(&ts, nullptr);
		if (ns < 400000)
			ns += ns / 4;
		q = hipEventQuery(ev);
	}
	return q;
}
inline hipError_t stream_wait(hipStream_t s)
{
	thread_local hipEvent_t ev = nullptr;
	if (!ev && hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) {
		ev = nullptr;
		return hipStreamSynchronize(s);
	}
	if (hipEventRecord(ev, s) != hipSuccess) { // e.g. the thread moved to another device since
		(void)hipGetLastError();
		return hipStreamSynchronize(s);
	}
	return event_wait(ev);
}
// Device -> pageable host memory (a stack variable, a std::vector).  The runtime stages such a copy and makes the
// calling thread WAIT ACTIVELY for everything queued before it on the stream -- behind a 400 ms resolver launch or a
// finder that is a whole core burning for that long (measured: 48 + 37 CPU-seconds per 16 GiB step in the scanner and
// GPU-worker threads, a fifth of the host's quota, taken from the LZMA parser).  So: sleep until the stream is idle,
// then copy.
inline hipError_t d2h_pageable(void *dst, const void *src, size_t bytes, hipStream_t s)
{
	hipError_t e = stream_wait(s);
	if (e != hipSuccess)
		return e;
	e = hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, s);
	if (e != hipSuccess)
		return e;
	return stream_wait(s);
}
int select_device(int device); // 0 or LRZGPU_E_*
int lzma_normalize(LzmaParams &p, int level, unsigned dictSize, int lc, int lp, int pb, int fb);
} // namespace lrzgpu
