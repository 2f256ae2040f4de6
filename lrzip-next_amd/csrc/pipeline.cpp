// pipeline.cpp -- what the pipeline's translation units share besides pipeline.h: the trace clock and the CPU seconds of
// its threads by role.
#include "pipeline.h"

namespace lrzgpu {

double g_trace_t0 = 0;
std::atomic<int> g_trace_events{0};

// CPU seconds burnt by the threads of the whole-file pipeline, by role, since the last lrzgpu_profile_reset()
static std::mutex g_role_mu;
static double g_role_cpu[8] = {0, 0, 0, 0, 0, 0, 0, 0};
void role_cpu_add(int role, double s)
{
	std::lock_guard<std::mutex> lk(g_role_mu);
	g_role_cpu[role & 7] += s;
}

} // namespace lrzgpu

// (lrzgpu_hash.h) CPU seconds of the pipeline's threads by role: 0 encoders (parser + range coder / zstd), 1 GPU workers
// (block copies, finder launches, list copies), 2 scanners, 3 the whole-input hash, 4 the reader; reset != 0 clears
extern "C" void lrzgpu_profile_cpu(double out[8], int reset)
{
	std::lock_guard<std::mutex> lk(lrzgpu::g_role_mu);
	for (int k = 0; k < 8; k++) {
		if (out)
			out[k] = lrzgpu::g_role_cpu[k];
		if (reset)
			lrzgpu::g_role_cpu[k] = 0;
	}
}
