// lat_bench.hip -- dependent-chain load latency of ONE wavefront on a table of 16-byte slots
// (developer tool: what does a round trip of the rzip resolver's table walk cost?)
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned long long u64;

template <int NLOAD, int MODE>
__global__ void __launch_bounds__(64) k_chase(const uint4 *tbl, u64 mask, int iters, int active, u64 *out)
{
	const int lane = threadIdx.x;
	u64 idx = (0x9E3779B97F4A7C15ull * (lane + 1)) & mask;
	u64 acc = 0;
	const u64 t0 = __builtin_amdgcn_s_memtime();
	for (int it = 0; it < iters; it++) {
		uint4 c[NLOAD];
		if (MODE == 0) { // every lane walks its own NLOAD consecutive slots
			if (lane < active) {
#pragma unroll
				for (int q = 0; q < NLOAD; q++)
					c[q] = tbl[idx + q];
			} else {
#pragma unroll
				for (int q = 0; q < NLOAD; q++)
					c[q] = make_uint4(0, 0, 0, 0);
			}
		} else { // transposed: instruction q serves lanes 4q..4q+3, 16 consecutive slots each (NLOAD == 16)
#pragma unroll
			for (int q = 0; q < NLOAD; q++) {
				const int owner = 4 * q + (lane >> 4);
				const u64 oidx = __shfl(idx, owner);
				if (owner < active)
					c[q] = tbl[oidx + (lane & 15)];
				else
					c[q] = make_uint4(0, 0, 0, 0);
			}
		}
		u64 h = 0;
#pragma unroll
		for (int q = 0; q < NLOAD; q++)
			h += c[q].x + c[q].z;
		if (MODE == 1) { // the owner needs something from every slot of its region
#pragma unroll
			for (int d = 1; d < 16; d <<= 1)
				h += __shfl_xor(h, d);
		}
		acc += h;
		idx = (idx * 6364136223846793005ull + h + 1442695040888963407ull) & mask;
	}
	const u64 t1 = __builtin_amdgcn_s_memtime();
	if (lane == 0)
		out[0] = t1 - t0;
	out[1 + lane] = acc;
}

int main(int argc, char **argv)
{
	const size_t mb = argc > 1 ? atoi(argv[1]) : 64;
	const size_t slots = mb * 65536;
	uint4 *tbl;
	hipMalloc(&tbl, (slots + 64) * 16);
	hipMemset(tbl, 0, (slots + 64) * 16);
	u64 *out;
	hipMalloc(&out, 65 * 8);
	const int iters = 20000;
	auto run = [&](const char *name, auto kern, int active) {
		hipLaunchKernelGGL(kern, dim3(1), dim3(64), 0, 0, tbl, (u64)(slots - 1), iters, active, out);
		hipDeviceSynchronize();
		hipEvent_t e0, e1;
		hipEventCreate(&e0);
		hipEventCreate(&e1);
		hipEventRecord(e0, 0);
		hipLaunchKernelGGL(kern, dim3(1), dim3(64), 0, 0, tbl, (u64)(slots - 1), iters, active, out);
		hipEventRecord(e1, 0);
		hipEventSynchronize(e1);
		float ms;
		hipEventElapsedTime(&ms, e0, e1);
		u64 cyc;
		hipMemcpy(&cyc, out, 8, hipMemcpyDeviceToHost);
		printf("%-28s active %2d: %7.0f clk/iter  %7.1f ns/iter\n", name, active, (double)cyc / iters, ms * 1e6 / iters);
	};
	printf("table %zu MiB\n", mb);
	for (int active : {1, 8, 35, 64}) {
		run("1 load/lane", k_chase<1, 0>, active);
		run("4 loads/lane", k_chase<4, 0>, active);
		run("8 loads/lane", k_chase<8, 0>, active);
		run("16 loads/lane", k_chase<16, 0>, active);
		run("32 loads/lane", k_chase<32, 0>, active);
		run("16 loads transposed", k_chase<16, 1>, active);
	}
	return 0;
}
