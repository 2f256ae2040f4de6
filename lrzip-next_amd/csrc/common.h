// common.h -- small helpers shared by the host-side translation units.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "lzma_enc.h"

namespace lrzgpu {
// Wait for a stream without burning a host core: hipStreamSynchronize() busy-waits, and the host
// cores are what the LZMA encoders need.  One blocking-sync event per calling thread.
inline hipError_t stream_wait(hipStream_t s)
{
	thread_local hipEvent_t ev = nullptr;
	if (!ev && hipEventCreateWithFlags(&ev, hipEventBlockingSync | hipEventDisableTiming) != hipSuccess) {
		ev = nullptr;
		return hipStreamSynchronize(s);
	}
	if (hipEventRecord(ev, s) != hipSuccess) { // e.g. the thread moved to another device since
		(void)hipGetLastError();
		return hipStreamSynchronize(s);
	}
	return hipEventSynchronize(ev);
}
int select_device(int device); // 0 or LRZGPU_E_*
int lzma_normalize(LzmaParams &p, int level, unsigned dictSize, int lc, int lp, int pb, int fb);
} // namespace lrzgpu
