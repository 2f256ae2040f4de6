// rzip_census.hip -- "does any 31-byte window of this chunk occur twice?", answered exactly, at HBM speed.
//
// A match of the rzip scan is at least MINIMUM_MATCH = 31 bytes long (single_match_len returns 0 below that,
// src/rzip.c:431-461), so a chunk in which no 31-byte window occurs twice cannot produce one whatever the hash table
// does: stream 0 is literal tokens, stream 1 is the input.  That is every chunk of incompressible data (BASELINE
// configs[4]: 32 GiB of random bytes), where the exact table automaton (k_resolve_mw, one workgroup per chunk) is the
// slowest thing in the library: 50 s for those 32 GiB against 34 s for their MD5.  The census replaces it there.
//
//   anchors   the value v[i] of the 8 bytes at every position i; position i is an ANCHOR if v[i] is the rightmost
//             minimum of v over some run of 24 consecutive positions that contains i (winnowing).  A 31-byte window holds
//             exactly 24 such 8-byte values, so it holds an anchor, and two equal windows hold it at the same relative
//             place, with the same value.  ~8 % of the positions of random data are anchors.
//   census    the anchors' values sorted (rocPRIM radix sort, keys only); two equal neighbours = two windows that MAY be
//             equal.  None: no 31-byte window occurs twice -- exactly, not probably.  Any: the answer is "maybe" and
//             the caller runs the resolver as ever (chance equality of two 8-byte values: 0.2 expected among the
//             2.7 G anchors of 32 GiB).
//   sample    first the same over the anchors whose mixed value ends in six zero bits (1/64 of them): ordinary data
//             shows its repeats there after a few milliseconds and pays nothing more.
// Bound: HBM, ~1 B read per position + 0.64 B of keys written and sorted per position.
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>

#include <cstdint>
#include <cstdio>

#include "common.h"
#include "pools.h"
#include "rzip_census.h"

namespace lrzgpu {
namespace {

constexpr int kSpan = 24;          // 8-byte values in a 31-byte window
constexpr int kReach = kSpan - 1;  // neighbours looked at on either side
constexpr int kTile = 2048;        // positions per workgroup
constexpr int kThreads = 256;

__device__ __forceinline__ uint64_t mix64(uint64_t x)
{
	x ^= x >> 31;
	x *= 0x9E3779B97F4A7C15ull;
	return x ^ (x >> 29);
}

// keys[] receives the value of every anchor in [0, npos) (positions with 8 bytes inside the chunk) that passes the
// sample mask; *count their number (keys beyond cap are dropped, the count still runs: the caller sees the overflow)
__global__ void __launch_bounds__(kThreads) k_winnow(const uint8_t *__restrict__ buf, int64_t n, int64_t npos, uint64_t sample_mask,
						     uint64_t *__restrict__ keys, unsigned long long cap, unsigned long long *__restrict__ count)
{
	// the tile's bytes (+ the neighbours' on both sides), then the 8-byte value at every position of it
	constexpr int kVals = kTile + 2 * kReach;
	constexpr int kPieces = (kVals + 7 + 15 + 15) / 16 + 1; // 16-byte pieces staged (the start is aligned down)
	__shared__ __attribute__((aligned(16))) uint32_t raw[kPieces * 4 + 4];
	__shared__ uint64_t val[kVals];
	const int64_t tile0 = (int64_t)blockIdx.x * kTile;
	const int64_t first = tile0 - kReach;                // position of val[0] (may be negative)
	const int64_t byte0 = first < 0 ? 0 : first & ~15ll; // aligned start of the bytes staged
	for (int w = threadIdx.x; w < kPieces; w += kThreads) {
		const int64_t at = byte0 + (int64_t)w * 16;
		uint4 q = make_uint4(0, 0, 0, 0);
		if (at + 16 <= n)
			q = *reinterpret_cast<const uint4 *>(buf + at);
		else if (at < n) {
			uint32_t t[4] = {0, 0, 0, 0};
			for (int k = 0; k < 16 && at + k < n; k++)
				t[k >> 2] |= (uint32_t)buf[at + k] << (8 * (k & 3));
			q = make_uint4(t[0], t[1], t[2], t[3]);
		}
		*reinterpret_cast<uint4 *>(&raw[w * 4]) = q;
	}
	__syncthreads();
	for (int k = threadIdx.x; k < kVals; k += kThreads) {
		const int64_t p = first + k;
		uint64_t v = ~0ull; // outside the chunk: never the smaller one
		if (p >= 0 && p < npos) {
			const int o = (int)(p - byte0);
			const uint32_t w0 = raw[o >> 2], w1 = raw[(o >> 2) + 1], w2 = raw[(o >> 2) + 2];
			const int sh = (o & 3) * 8;
			v = (uint64_t)__funnelshift_r(w0, w1, sh) | (uint64_t)__funnelshift_r(w1, w2, sh) << 32;
		}
		val[k] = v;
	}
	__syncthreads();
	const int lane = threadIdx.x & 63;
	for (int k = threadIdx.x; k < kTile; k += kThreads) { // (kTile is a multiple of kThreads: whole wavefronts all the way)
		const int64_t p = tile0 + k;
		bool anchor = false;
		uint64_t v = 0;
		if (p < npos) {
			const int c = k + kReach;
			v = val[c];
			// a: positions before it, without a gap, whose value is not smaller (a tie goes to the later position);
			// b: positions after it whose value is greater.  It is the rightmost minimum of some run of 24 <=> a + b >= 23.
			int a = 0, b = 0;
			while (a < kReach && (p - 1 - a < 0 || val[c - 1 - a] >= v))
				a++;
			if (a < kReach) {
				const int need = kReach - a;
				while (b < need && (p + 1 + b >= npos || val[c + 1 + b] > v))
					b++;
			}
			anchor = a + b >= kReach && (mix64(v) & sample_mask) == 0;
		}
		const unsigned long long m = __ballot(anchor);
		if (m) {
			unsigned long long base = 0;
			if (lane == 0)
				base = atomicAdd(count, (unsigned long long)__popcll(m));
			base = __shfl(base, 0);
			if (anchor) {
				const unsigned long long at = base + (unsigned long long)__popcll(m & ((1ull << lane) - 1));
				if (at < cap)
					keys[at] = v;
			}
		}
	}
}

__global__ void __launch_bounds__(256) k_equal_neighbours(const uint64_t *__restrict__ keys, unsigned long long n, unsigned long long *__restrict__ found)
{
	const unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
	const bool eq = i + 1 < n && keys[i] == keys[i + 1];
	const unsigned long long m = __ballot(eq);
	if (m && (threadIdx.x & 63) == 0)
		atomicAdd(found, (unsigned long long)__popcll(m));
}

// one pass: anchors under `sample_mask` -> sorted -> equal neighbours.  *dups = their number, *anchors = anchors seen;
// returns 0, or 1 when the anchors did not fit `cap_keys` (then nothing is known), or a negative error
int census_pass(const uint8_t *d, int64_t n, uint64_t sample_mask, unsigned long long cap_keys, int device, hipStream_t s, unsigned long long *dups,
		unsigned long long *anchors)
{
	const int64_t npos = n - 7;
	DevBuf keys_a, keys_b, tmp, scal;
	if (!keys_a.alloc((size_t)cap_keys * 8 + 64, device) || !keys_b.alloc((size_t)cap_keys * 8 + 64, device) || !scal.alloc(64, device))
		return -2;
	unsigned long long *d_count = (unsigned long long *)scal.p, *d_found = d_count + 1;
	if (hipMemsetAsync(scal.p, 0, 64, s) != hipSuccess)
		return -1;
	const int64_t tiles = (npos + kTile - 1) / kTile;
	hipLaunchKernelGGL(k_winnow, dim3((unsigned)tiles), dim3(kThreads), 0, s, d, n, npos, sample_mask, (uint64_t *)keys_a.p, cap_keys, d_count);
	unsigned long long h[2] = {0, 0};
	if (d2h_pageable(h, scal.p, 16, s) != hipSuccess)
		return -1;
	*anchors = h[0];
	*dups = 0;
	if (h[0] > cap_keys)
		return 1;
	if (h[0] < 2)
		return 0;
	size_t tb = 0;
	if (rocprim::radix_sort_keys(nullptr, tb, (uint64_t *)keys_a.p, (uint64_t *)keys_b.p, (size_t)h[0], 0, 64, s) != hipSuccess)
		return -1;
	if (!tmp.alloc(tb + 256, device))
		return -2;
	if (rocprim::radix_sort_keys(tmp.p, tb, (uint64_t *)keys_a.p, (uint64_t *)keys_b.p, (size_t)h[0], 0, 64, s) != hipSuccess)
		return -1;
	hipLaunchKernelGGL(k_equal_neighbours, dim3((unsigned)((h[0] + 255) / 256)), dim3(256), 0, s, (const uint64_t *)keys_b.p, h[0], d_found);
	if (d2h_pageable(h, scal.p, 16, s) != hipSuccess)
		return -1;
	*dups = h[1];
	return 0;
}

} // namespace

int duplicate_census(const uint8_t *d_chunk, int64_t n, int device, hipStream_t s, CensusStats *st)
{
	CensusStats local;
	if (!st)
		st = &local;
	*st = CensusStats();
	if (n < 31)
		return 1; // no 31-byte window at all
	if (n - 7 > (int64_t)0x7FFFFFFF * kTile)
		return 0;
	unsigned long long dups = 0, anchors = 0;
	// the sample: 1/64 of the anchors (those of ordinary data repeat: done after a few milliseconds)
	const unsigned long long cap_s = (unsigned long long)(n / 256) + 65536;
	int r = census_pass(d_chunk, n, 63, cap_s, device, s, &dups, &anchors);
	if (r < 0)
		return r;
	st->sample_anchors = (int64_t)anchors;
	st->sample_equal = (int64_t)dups;
	if (r == 1 || dups)
		return 0;
	// all of them
	const unsigned long long cap = (unsigned long long)(n / 8) + 65536;
	r = census_pass(d_chunk, n, 0, cap, device, s, &dups, &anchors);
	if (r < 0)
		return r == -2 ? 0 : r; // (no room for the keys: the resolver it is)
	st->anchors = (int64_t)anchors;
	st->equal = (int64_t)dups;
	return r == 0 && dups == 0 ? 1 : 0;
}

} // namespace lrzgpu
