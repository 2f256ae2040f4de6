// lz4_gate.hip -- the lz4 compressibility gate of the stream layer on the GPU (gfx950).
//
// Reference: src/stream.c:2325-2380 lz4_compresses() calls liblz4's LZ4_compress_default() only to
// learn the compressed SIZE of a stream block.  The decision must equal liblz4's bit for bit, so
// the kernel is an exact size-only emulation of liblz4 1.9.3 LZ4_compress_generic (acceleration 1,
// limitedOutput, noDict; byU16 hash below 64 KB + 11, otherwise byU32 with the 5-byte hash of
// 64-bit little-endian hosts).  The greedy scan is a serial automaton per block (hash table state
// + skip acceleration), so the parallelism is ACROSS stream blocks: one wavefront per block, hash
// table in LDS (32 KiB), many blocks per launch.  Lane 0 drives the automaton; all 64 lanes join
// the match-length extension (512 B per step) through wave ballots.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "lz4_gate.h"

namespace lrzgpu {

namespace {
constexpr int MINMATCH = 4;
constexpr int LASTLITERALS = 5;
constexpr int MFLIMIT = 12;
constexpr int LZ4_MINLENGTH = MFLIMIT + 1;
constexpr int LZ4_64KLIMIT = 65536 + (MFLIMIT - 1);
constexpr int LZ4_SKIPTRIGGER = 6;
constexpr int LZ4_HASHLOG = 12;
constexpr unsigned ML_MASK = 15, RUN_MASK = 15;
constexpr unsigned LZ4_DISTANCE_MAX = 65535;
constexpr unsigned LZ4_MAX_INPUT_SIZE = 0x7E000000;

__device__ __forceinline__ uint32_t rd32(const uint8_t *p)
{
	return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}
__device__ __forceinline__ uint64_t rd64(const uint8_t *p)
{
	return (uint64_t)rd32(p) | ((uint64_t)rd32(p + 4) << 32);
}
__device__ __forceinline__ uint32_t hash_pos(const uint8_t *p, bool by_u16)
{
	if (by_u16)
		return (rd32(p) * 2654435761U) >> (MINMATCH * 8 - (LZ4_HASHLOG + 1));
	return (uint32_t)(((rd64(p) << 24) * 889523592379ULL) >> (64 - LZ4_HASHLOG));
}

// Wave-cooperative LZ4_count: number of equal bytes of a[] and b[] before alimit.
// Called by all 64 lanes with identical (wave-uniform) arguments.
__device__ __forceinline__ uint32_t wave_count_eq(const uint8_t *a, const uint8_t *b, const uint8_t *alimit, int lane)
{
	uint32_t done = 0;
	const uint32_t total = (uint32_t)(alimit - a);
	for (;;) {
		// each lane checks 8 bytes
		uint32_t off = done + (uint32_t)lane * 8;
		uint32_t mism = 8; // first mismatching byte within my 8, 8 = none
		if (off < total) {
			uint32_t lim = total - off < 8 ? total - off : 8;
			mism = lim;
			for (uint32_t k = 0; k < lim; k++)
				if (a[off + k] != b[off + k]) {
					mism = k;
					break;
				}
			if (mism == lim && lim == 8)
				mism = 8;
		} else
			mism = 0; // beyond the limit: acts as a stop
		unsigned long long stop = __ballot(mism < 8);
		if (stop) {
			int first = __ffsll((long long)stop) - 1;
			uint32_t m = __shfl(mism, first);
			uint32_t res = done + (uint32_t)first * 8 + m;
			return res < total ? res : total;
		}
		done += 512;
		if (done >= total)
			return total;
	}
}
} // namespace

// grid.x = number of jobs, block = 64 threads (one wavefront per job)
__global__ void __launch_bounds__(64) k_lz4_size(const Lz4Job *__restrict__ jobs, int *__restrict__ results)
{
	__shared__ uint32_t table[1 << (LZ4_HASHLOG + 1)];
	const int lane = threadIdx.x;
	const Lz4Job job = jobs[blockIdx.x];
	const uint8_t *src = job.src;
	const int src_size = job.src_size;
	const long long olimit = job.dst_capacity;

	if ((uint32_t)src_size > LZ4_MAX_INPUT_SIZE) {
		if (lane == 0)
			results[blockIdx.x] = 0;
		return;
	}
	if (src_size == 0) {
		if (lane == 0)
			results[blockIdx.x] = job.dst_capacity > 0 ? 1 : 0;
		return;
	}
	for (int k = lane; k < (1 << (LZ4_HASHLOG + 1)); k += 64)
		table[k] = 0;
	__syncthreads();

	const bool limited = job.dst_capacity < (long long)src_size + src_size / 255 + 16;
	const bool by_u16 = src_size < LZ4_64KLIMIT;
	const uint8_t *const iend = src + src_size;
	const uint8_t *const mflimit_plus_one = iend - MFLIMIT + 1;
	const uint8_t *const matchlimit = iend - LASTLITERALS;

	// wave-uniform automaton state (every lane carries a copy; lane 0's LDS traffic is the only one)
	const uint8_t *ip = src, *anchor = src, *match = src;
	long long op = 0;
	int result = -1; // -1 = still running
	uint32_t forward_h = 0;
	// phases: 0 = search, 1 = encode match at (ip, match) (token already counted)
	if (src_size < LZ4_MINLENGTH)
		goto last_literals;

	if (lane == 0)
		table[hash_pos(ip, by_u16)] = 0;
	ip++;
	forward_h = hash_pos(ip, by_u16);

	for (;;) {
		// ---- find a match: serial greedy scan (lane 0 computes, result broadcast) ----
		{
			int found = 0; // 1 = match, 2 = hit end
			const uint8_t *r_ip = ip, *r_match = match;
			uint32_t r_fh = forward_h;
			if (lane == 0) {
				const uint8_t *forward_ip = ip;
				int step = 1;
				int search_nb = 1 << LZ4_SKIPTRIGGER;
				for (;;) {
					uint32_t h = r_fh;
					uint32_t current = (uint32_t)(forward_ip - src);
					uint32_t match_index = table[h];
					r_ip = forward_ip;
					forward_ip += step;
					step = search_nb++ >> LZ4_SKIPTRIGGER;
					if (forward_ip > mflimit_plus_one) {
						found = 2;
						break;
					}
					r_match = src + match_index;
					r_fh = hash_pos(forward_ip, by_u16);
					table[h] = current;
					if (!by_u16 && match_index + LZ4_DISTANCE_MAX < current)
						continue;
					if (rd32(r_match) == rd32(r_ip)) {
						found = 1;
						break;
					}
				}
				// catch up
				if (found == 1)
					while (r_ip > anchor && r_match > src && r_ip[-1] == r_match[-1]) {
						r_ip--;
						r_match--;
					}
			}
			found = __shfl(found, 0);
			{
				unsigned long long a = (unsigned long long)r_ip, b = (unsigned long long)r_match;
				a = ((unsigned long long)__shfl((unsigned)(a >> 32), 0) << 32) | (unsigned)__shfl((unsigned)a, 0);
				b = ((unsigned long long)__shfl((unsigned)(b >> 32), 0) << 32) | (unsigned)__shfl((unsigned)b, 0);
				ip = (const uint8_t *)a;
				match = (const uint8_t *)b;
			}
			forward_h = __shfl(r_fh, 0);
			if (found == 2)
				goto last_literals;
		}
		// ---- literals ----
		{
			unsigned lit = (unsigned)(ip - anchor);
			op++; // token
			if (limited && op + lit + (2 + 1 + LASTLITERALS) + (lit / 255) > olimit) {
				result = 0;
				goto done;
			}
			if (lit >= RUN_MASK)
				op += (lit - RUN_MASK) / 255 + 1;
			op += lit;
		}
	next_match:
		op += 2; // offset
		{
			unsigned mc = wave_count_eq(ip + MINMATCH, match + MINMATCH, matchlimit, lane);
			ip += (size_t)mc + MINMATCH;
			if (limited && op + (1 + LASTLITERALS) + (mc + 240) / 255 > olimit) {
				result = 0;
				goto done;
			}
			if (mc >= ML_MASK)
				op += (mc - ML_MASK) / 255 + 1;
		}
		anchor = ip;
		if (ip >= mflimit_plus_one)
			break;
		{
			int again = 0;
			uint32_t r_midx = 0;
			if (lane == 0) {
				table[hash_pos(ip - 2, by_u16)] = (uint32_t)(ip - 2 - src);
				uint32_t h = hash_pos(ip, by_u16);
				uint32_t current = (uint32_t)(ip - src);
				uint32_t match_index = table[h];
				table[h] = current;
				r_midx = match_index;
				if ((by_u16 || match_index + LZ4_DISTANCE_MAX >= current) && rd32(src + match_index) == rd32(ip))
					again = 1;
			}
			again = __shfl(again, 0);
			r_midx = __shfl(r_midx, 0);
			if (again) {
				match = src + r_midx;
				op++; // token, zero literals
				goto next_match;
			}
		}
		ip++;
		forward_h = hash_pos(ip, by_u16);
	}

last_literals:
	{
		long long last_run = (long long)(iend - anchor);
		if (limited && op + last_run + 1 + ((last_run + 255 - RUN_MASK) / 255) > olimit) {
			result = 0;
			goto done;
		}
		if (last_run >= (long long)RUN_MASK)
			op += 1 + (last_run - RUN_MASK) / 255 + 1;
		else
			op += 1;
		op += last_run;
		result = (int)op;
	}
done:
	if (lane == 0)
		results[blockIdx.x] = result;
}

int lz4_sizes_device(const Lz4Job *d_jobs, int njobs, int *d_results, hipStream_t s)
{
	if (njobs <= 0)
		return 0;
	hipLaunchKernelGGL(k_lz4_size, dim3(njobs), dim3(64), 0, s, d_jobs, d_results);
	return hipGetLastError() == hipSuccess ? 0 : -1;
}

} // namespace lrzgpu
