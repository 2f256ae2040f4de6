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
//   census    the anchors' values sorted (rocPRIM radix sort, keys only) and every value compared with its successors
//             of equal upper half; two equal values = two windows that MAY be equal.  None: no 31-byte window occurs twice -- exactly, not probably.  A few (chance equality
//             of two 8-byte values: 0.2 expected among the 2.7 G anchors of 32 GiB): a second pass writes the positions
//             of the anchors with those values and the bytes around them decide.  Many: "maybe", the caller runs the
//             resolver as ever.
//   sample    first the same over the anchors whose mixed value ends in six zero bits (1/64 of them): ordinary data
//             shows its repeats there after a few milliseconds and pays nothing more.
// Bound: HBM, ~1 B read per position + 0.64 B of keys written and sorted per position.
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <ctime>
#include <vector>

#include "common.h"
#include "pools.h"
#include "rzip_census.h"

namespace lrzgpu {
namespace {

inline double now_s()
{
	timespec t;
	clock_gettime(CLOCK_MONOTONIC, &t);
	return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}

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
// With `wanted` (a short list of values): keys[] receives POSITIONS instead, those of the anchors whose value is in the list.
constexpr int kMaxSuspects = 64;
__global__ void __launch_bounds__(kThreads) k_winnow(const uint8_t *__restrict__ buf, int64_t n, int64_t npos, int64_t tile_first, uint64_t sample_mask,
						     uint64_t *__restrict__ keys, unsigned long long cap, unsigned long long *__restrict__ count,
						     const uint64_t *__restrict__ wanted, int n_wanted)
{
	// the tile's bytes (+ the neighbours' on both sides), then the 8-byte value at every position of it
	constexpr int kVals = kTile + 2 * kReach;
	constexpr int kPieces = (kVals + 7 + 15 + 15) / 16 + 1; // 16-byte pieces staged (the start is aligned down)
	__shared__ __attribute__((aligned(16))) uint32_t raw[kPieces * 4 + 4];
	__shared__ uint64_t val[kVals];
	const int64_t tile0 = (tile_first + (int64_t)blockIdx.x) * kTile;
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
	// every thread's eight positions first, then ONE reservation of output space for the whole tile (a reservation per
	// wavefront and round was 134 M atomic additions to one address for 8 GiB: 1.4 of the pass's 1.6 s)
	constexpr int kPer = kTile / kThreads;
	__shared__ unsigned int wave_total[kThreads / 64];
	__shared__ unsigned long long tile_base;
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	uint64_t mine[kPer];
	unsigned int flags = 0; // bit j: this thread's j-th position is an anchor
	unsigned int rank_of[kPer]; // ... and its place among the wavefront's anchors
	unsigned int wave_count = 0;
#pragma unroll
	for (int j = 0; j < kPer; j++) {
		const int k = threadIdx.x + j * kThreads;
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
			if (anchor && n_wanted) {
				bool in_list = false;
				for (int q = 0; q < n_wanted; q++)
					in_list |= wanted[q] == v;
				anchor = in_list;
				v = (uint64_t)p;
			}
		}
		mine[j] = v;
		const unsigned long long m = __ballot(anchor);
		if (anchor)
			flags |= 1u << j;
		// this lane's slot among the wavefront's anchors so far
		rank_of[j] = wave_count + (unsigned int)__popcll(m & ((1ull << lane) - 1));
		wave_count += (unsigned int)__popcll(m);
	}
	if (lane == 0)
		wave_total[wave] = wave_count;
	__syncthreads();
	if (threadIdx.x == 0) {
		unsigned int total = 0;
		for (int w = 0; w < kThreads / 64; w++)
			total += wave_total[w];
		tile_base = total ? atomicAdd(count, (unsigned long long)total) : 0;
	}
	__syncthreads();
	unsigned long long base = tile_base;
	for (int w = 0; w < wave; w++)
		base += wave_total[w];
#pragma unroll
	for (int j = 0; j < kPer; j++)
		if (flags >> j & 1) {
			const unsigned long long at = base + rank_of[j];
			if (at < cap)
				keys[at] = mine[j];
		}
}

// keys sorted: every key looks at its successors of equal upper half for an equal one (its neighbour, in fact); the values
// found go to a short list (the first kMaxSuspects), their number to *found.
__global__ void __launch_bounds__(256) k_equal_in_runs_slice(const uint64_t *__restrict__ keys, unsigned long long first, unsigned long long count,
							     unsigned long long n, unsigned long long *__restrict__ found, uint64_t *__restrict__ values)
{
	const unsigned long long t = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
	if (t >= count)
		return;
	const unsigned long long i = first + t;
	const uint64_t k = keys[i];
	for (unsigned long long j = i + 1; j < n; j++) {
		const uint64_t o = keys[j];
		if ((o >> 32) != (k >> 32))
			break;
		if (o == k) {
			const unsigned long long at = atomicAdd(found, 1ull);
			if (at < (unsigned long long)kMaxSuspects)
				values[at] = k;
			break; // (one report per key: a value that occurs m times is reported m - 1 times)
		}
	}
}

// one pass: anchors under `sample_mask` -> sorted -> equal values.  *dups = equal pairs found, suspects[0 .. min(*dups,
// kMaxSuspects)) their values, *anchors = anchors seen; returns 0, or 1 when nothing can be said (the anchors did not fit
// `cap_keys`), or a negative error
int launch_winnow(const uint8_t *d, int64_t n, uint64_t sample_mask, uint64_t *keys, unsigned long long cap, unsigned long long *d_count,
		  const uint64_t *wanted, int n_wanted, hipStream_t s)
{
	const int64_t npos = n - 7;
	const int64_t tiles = (npos + kTile - 1) / kTile;
	constexpr int64_t kLaunchTiles = 1 << 20; // (a launch of more than 2^32 threads is refused: 32 GiB would be exactly that)
	(void)hipGetLastError();
	for (int64_t t0 = 0; t0 < tiles; t0 += kLaunchTiles) {
		const int64_t nt = tiles - t0 < kLaunchTiles ? tiles - t0 : kLaunchTiles;
		hipLaunchKernelGGL(k_winnow, dim3((unsigned)nt), dim3(kThreads), 0, s, d, n, npos, t0, sample_mask, keys, cap, d_count, wanted, n_wanted);
		if (hipGetLastError() != hipSuccess)
			return -1;
	}
	return 0;
}
int census_pass(const uint8_t *d, int64_t n, uint64_t sample_mask, unsigned long long cap_keys, int device, hipStream_t s, unsigned long long *dups,
		unsigned long long *anchors, uint64_t *suspects)
{
	const int64_t npos = n - 7;
	DevBuf keys_a, keys_b, tmp, scal;
	if (!keys_a.alloc((size_t)cap_keys * 8 + 64, device) || !keys_b.alloc((size_t)cap_keys * 8 + 64, device) || !scal.alloc(64 + kMaxSuspects * 8, device))
		return -2;
	unsigned long long *d_count = (unsigned long long *)scal.p, *d_found = d_count + 1;
	uint64_t *d_values = (uint64_t *)(scal.p + 64);
	if (hipMemsetAsync(scal.p, 0, 64 + kMaxSuspects * 8, s) != hipSuccess)
		return -1;
	const bool trace = getenv("LRZGPU_TRACE") != nullptr;
	const double t_a = trace ? now_s() : 0;
	if (launch_winnow(d, n, sample_mask, (uint64_t *)keys_a.p, cap_keys, d_count, nullptr, 0, s) != 0)
		return -1;
	unsigned long long h[2] = {0, 0};
	if (d2h_pageable(h, scal.p, 16, s) != hipSuccess)
		return -1;
	const double t_b = trace ? now_s() : 0;
	*anchors = h[0];
	*dups = 0;
	if (h[0] > cap_keys)
		return 1;
	// every run of 24 positions holds an anchor: a full pass that found fewer than that was no pass (nothing may be
	// concluded from keys that are not there)
	if (sample_mask == 0 && (int64_t)h[0] < npos / kSpan - 1)
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
	for (unsigned long long k0 = 0; k0 < h[0]; k0 += 1ull << 30) { // (again: launches below 2^32 threads)
		// a slice looks beyond its end along the last run: the keys behind it are there (n counts from the slice's start)
		const unsigned long long nk = h[0] - k0 < (1ull << 30) ? h[0] - k0 : (1ull << 30);
		hipLaunchKernelGGL(k_equal_in_runs_slice, dim3((unsigned)((nk + 255) / 256)), dim3(256), 0, s, (const uint64_t *)keys_b.p, k0, nk, h[0], d_found, d_values);
		if (hipGetLastError() != hipSuccess)
			return -1;
	}
	if (d2h_pageable(h, scal.p, 16, s) != hipSuccess)
		return -1;
	if (trace)
		fprintf(stderr, "lrzgpu census: %lld bytes, mask %llx: anchors %.1f ms (%llu), sort + compare %.1f ms\n", (long long)n, (unsigned long long)sample_mask,
			(t_b - t_a) * 1e3, *anchors, (now_s() - t_b) * 1e3);
	*dups = h[1];
	if (h[1] && suspects && d2h_pageable(suspects, scal.p + 64, kMaxSuspects * 8, s) != hipSuccess)
		return -1;
	return 0;
}

// The few values that two anchors share: do two 31-byte windows share them too?  The anchors' positions come out of a
// second pass (only the values in question are written); two windows that are equal hold their anchor at the same
// relative place, so positions A and B belong to equal windows iff the bytes before them agree for l and the bytes
// from them on for r bytes with l <= 23, r <= 31 and l + r >= 31.  1: some pair does (or there are too many positions
// to say), 0: none does, < 0: error.
int verify_suspects(const uint8_t *d, int64_t n, const uint64_t *values, int nv, int device, hipStream_t s)
{
	constexpr unsigned long long kMaxPos = 4096;
	DevBuf pos, scal, want;
	if (!pos.alloc(kMaxPos * 8, device) || !scal.alloc(64, device) || !want.alloc((size_t)kMaxSuspects * 8, device))
		return -2;
	if (hipMemsetAsync(scal.p, 0, 64, s) != hipSuccess || hipMemcpyAsync(want.p, values, (size_t)nv * 8, hipMemcpyHostToDevice, s) != hipSuccess)
		return -1;
	if (launch_winnow(d, n, 0, (uint64_t *)pos.p, kMaxPos, (unsigned long long *)scal.p, (const uint64_t *)want.p, nv, s) != 0)
		return -1;
	unsigned long long cnt = 0;
	if (d2h_pageable(&cnt, scal.p, 8, s) != hipSuccess)
		return -1;
	if (cnt > kMaxPos)
		return 1;
	std::vector<uint64_t> ps((size_t)cnt);
	if (cnt && d2h_pageable(ps.data(), pos.p, (size_t)cnt * 8, s) != hipSuccess)
		return -1;
	// the bytes around every position: 23 before, 31 from it on (what lies outside the chunk agrees with nothing)
	struct Around {
		int64_t p;
		uint8_t b[54];
		int lo, hi; // valid bytes: b[lo .. hi)
	};
	std::vector<Around> ar((size_t)cnt);
	for (size_t k = 0; k < ps.size(); k++) {
		Around &a = ar[k];
		a.p = (int64_t)ps[k];
		const int64_t from = a.p - 23 < 0 ? 0 : a.p - 23, to = a.p + 31 > n ? n : a.p + 31;
		a.lo = (int)(from - (a.p - 23));
		a.hi = (int)(to - (a.p - 23));
		if (d2h_pageable(a.b + a.lo, d + from, (size_t)(to - from), s) != hipSuccess)
			return -1;
	}
	for (size_t x = 0; x < ar.size(); x++)
		for (size_t y = x + 1; y < ar.size(); y++) {
			const Around &A = ar[x], &B = ar[y];
			if (memcmp(A.b + 23, B.b + 23, 8) != 0 || A.hi < 31 || B.hi < 31)
				continue; // (different values)
			int l = 0, r = 0;
			while (l < 23 && 22 - l >= A.lo && 22 - l >= B.lo && A.b[22 - l] == B.b[22 - l])
				l++;
			while (r < 31 && 23 + r < A.hi && 23 + r < B.hi && A.b[23 + r] == B.b[23 + r])
				r++;
			if (l + r >= 31)
				return 1;
		}
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
	int r = census_pass(d_chunk, n, 63, cap_s, device, s, &dups, &anchors, nullptr);
	if (r < 0)
		return r == -2 ? 0 : r;
	st->sample_anchors = (int64_t)anchors;
	st->sample_equal = (int64_t)dups;
	if (r == 1 || dups > (unsigned long long)kMaxSuspects)
		return 0; // (a few equal values may be chance: the full pass looks at them)
	// all of them -- two key arrays of 8 B per anchor (one per 8 positions at most) and the sort's own temporary: about
	// 3 n bytes for the moment, while other scanners and the finders of the run allocate beside this one.  A census that
	// would take the device below its margin is skipped (the resolver runs as ever) rather than make somebody else's
	// allocation fail
	if (DeviceBudget::free_now() < (size_t)n * 3 + DeviceBudget::margin())
		return 0;
	const unsigned long long cap = (unsigned long long)(n / 8) + 65536;
	uint64_t suspects[kMaxSuspects];
	r = census_pass(d_chunk, n, 0, cap, device, s, &dups, &anchors, suspects);
	if (r < 0)
		return r == -2 ? 0 : r; // (no room for the keys: the resolver it is)
	st->anchors = (int64_t)anchors;
	st->equal = (int64_t)dups;
	if (r != 0)
		return 0;
	if (dups == 0)
		return 1;
	if (dups > (unsigned long long)kMaxSuspects)
		return 0; // (that many equal values are no accident)
	// a handful of equal 8-byte values among billions: chance (0.2 expected in 32 GiB of noise) -- or a repeat; looked at
	const int v = verify_suspects(d_chunk, n, suspects, (int)dups, device, s);
	if (v < 0)
		return v == -2 ? 0 : v;
	st->cleared = v == 0 ? (int64_t)dups : 0;
	return v == 0 ? 1 : 0;
}

} // namespace lrzgpu
