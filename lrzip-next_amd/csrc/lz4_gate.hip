// lz4_gate.hip -- the lz4 compressibility gate of the stream layer on the GPU (gfx950).
//
// Reference: src/stream.c:2325-2380 lz4_compresses() calls liblz4's LZ4_compress_default() only to
// learn the compressed SIZE of a stream block.  The decision must equal liblz4's bit for bit, so
// the kernel is an exact size-only emulation of liblz4 1.9.3 LZ4_compress_generic (acceleration 1,
// limitedOutput, noDict; byU16 hash below 64 KB + 11, otherwise byU32 with the 5-byte hash of
// 64-bit little-endian hosts).  The greedy scan is a serial automaton per block (hash table state
// + skip acceleration), so the parallelism is ACROSS stream blocks: one wavefront per block, hash
// table (16 KiB) and a 128 KiB sliding window of the input in LDS, many blocks per launch.  Lane 0 drives the automaton; all 64 lanes join
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

} // namespace

// ---- LDS-resident sliding window -------------------------------------------------------------
// The automaton's loads (8 B at the scan position, 4 B at the candidate <= 64 KiB behind) are
// dependent and tiny; served from HBM/L2 they cost ~1.5k cycles per step.  All 64 lanes therefore
// stream the input into a 128 KiB LDS ring (coalesced 16-byte loads, 16 KiB at a time) that always
// holds the 64 KiB behind and ahead of the scan position; lane 0 reads the ring (~100 cycles).
// The ring is a cache only: any access outside it falls back to global memory, so correctness
// never depends on the window bookkeeping.
constexpr int RING_BITS = 17;
constexpr uint32_t RING = 1u << RING_BITS;
constexpr uint32_t RING_CHUNK = 16384;
constexpr uint32_t TABLE_BYTES = 16384; // 4096 x u32 (byU32) or 8192 x u16 (byU16)
constexpr uint32_t OWNER_BYTES = 8192;  // one lane id per hash value: same-hash probes of a round

struct Win {
	const uint8_t *src;
	const uint32_t *ring32;
	uint32_t lo, hi; // [lo, hi) of the input is resident in the ring

	__device__ __forceinline__ uint32_t rd32(uint32_t pos) const
	{
		if (pos >= lo && pos + 4 <= hi) {
			const uint32_t idx = pos & (RING - 1), a = idx >> 2, sh = (idx & 3) * 8;
			const uint32_t w0 = ring32[a], w1 = ring32[(a + 1) & (RING / 4 - 1)];
			return sh ? (w0 >> sh) | (w1 << (32 - sh)) : w0;
		}
		const uint8_t *p = src + pos;
		return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
	}
	__device__ __forceinline__ uint64_t rd64(uint32_t pos) const
	{
		if (pos >= lo && pos + 8 <= hi) {
			const uint32_t idx = pos & (RING - 1), a = idx >> 2, sh = (idx & 3) * 8;
			const uint32_t w0 = ring32[a], w1 = ring32[(a + 1) & (RING / 4 - 1)], w2 = ring32[(a + 2) & (RING / 4 - 1)];
			const uint32_t l = sh ? (w0 >> sh) | (w1 << (32 - sh)) : w0;
			const uint32_t h = sh ? (w1 >> sh) | (w2 << (32 - sh)) : w1;
			return (uint64_t)l | ((uint64_t)h << 32);
		}
		return (uint64_t)rd32(pos) | ((uint64_t)rd32(pos + 4) << 32);
	}
	__device__ __forceinline__ uint32_t rd8(uint32_t pos) const
	{
		if (pos >= lo && pos < hi) {
			const uint32_t idx = pos & (RING - 1);
			return (ring32[idx >> 2] >> ((idx & 3) * 8)) & 0xFF;
		}
		return src[pos];
	}
	// Wave-cooperative LZ4_count: equal bytes of the input at a.. and b.. before `alimit`
	// (all 64 lanes, wave-uniform arguments, 512 B per step, served from the ring)
	__device__ __forceinline__ uint32_t count_eq(uint32_t a, uint32_t b, uint32_t alimit, int lane) const
	{
		uint32_t done = 0;
		const uint32_t total = alimit - a;
		for (;;) {
			const uint32_t off = done + (uint32_t)lane * 8;
			uint32_t mism = 0; // beyond the limit: acts as a stop
			if (off < total) {
				const uint32_t lim = total - off < 8 ? total - off : 8;
				if (lim == 8) {
					const uint64_t x = rd64(a + off) ^ rd64(b + off);
					mism = x ? (uint32_t)(__ffsll((long long)x) - 1) >> 3 : 8;
				} else {
					mism = lim;
					for (uint32_t k = 0; k < lim; k++)
						if (rd8(a + off + k) != rd8(b + off + k)) {
							mism = k;
							break;
						}
				}
			}
			const unsigned long long stop = __ballot(mism < 8);
			if (stop) {
				const int first = __ffsll((long long)stop) - 1;
				const uint32_t res = done + (uint32_t)first * 8 + __shfl(mism, first);
				return res < total ? res : total;
			}
			done += 512;
			if (done >= total)
				return total;
		}
	}
	__device__ __forceinline__ uint32_t hash(uint32_t pos, bool by_u16) const
	{
		if (by_u16)
			return (rd32(pos) * 2654435761U) >> (MINMATCH * 8 - (LZ4_HASHLOG + 1));
		return (uint32_t)(((rd64(pos) << 24) * 889523592379ULL) >> (64 - LZ4_HASHLOG));
	}
};

// grid.x = number of jobs, block = 64 threads (one wavefront per job); dynamic LDS = RING + TABLE_BYTES + OWNER_BYTES
__global__ void __launch_bounds__(64) k_lz4_size(const Lz4Job *__restrict__ jobs, int *__restrict__ results)
{
	extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
	uint8_t *ring = smem;
	uint32_t *table32 = reinterpret_cast<uint32_t *>(smem + RING);
	uint16_t *table16 = reinterpret_cast<uint16_t *>(smem + RING);
	uint8_t *owner = smem + RING + TABLE_BYTES;
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
	for (int k = lane; k < (int)(TABLE_BYTES / 4); k += 64)
		table32[k] = 0;

	const bool limited = job.dst_capacity < (long long)src_size + src_size / 255 + 16;
	const bool by_u16 = src_size < LZ4_64KLIMIT;
	const uint32_t iend = (uint32_t)src_size;
	const uint32_t mflimit_plus_one = iend - MFLIMIT + 1; // only used when src_size >= LZ4_MINLENGTH
	const uint32_t matchlimit = iend - LASTLITERALS;
	const uint32_t load_end = (iend + 15u) & ~15u; // the job buffer is readable up to here (16 B pad)
	const bool aligned16 = (((uintptr_t)src) & 15) == 0;

	Win W;
	W.src = src;
	W.ring32 = reinterpret_cast<const uint32_t *>(ring);
	W.lo = W.hi = 0;
	// make [.., target) resident, target = 64 KiB ahead of `pos`, in 16 KiB steps
	auto ensure = [&](uint32_t pos) {
		uint64_t want = ((uint64_t)pos + 65536u) & ~(uint64_t)(RING_CHUNK - 1);
		uint32_t target = want > load_end ? load_end : (uint32_t)want;
		if (target <= W.hi)
			return;
		uint32_t from = W.hi;
		if (target - from > RING)
			from = (target - RING + 15u) & ~15u;
		for (uint32_t off = from + (uint32_t)lane * 16; off < target; off += 64 * 16) {
			uint4 v;
			if (aligned16)
				v = *reinterpret_cast<const uint4 *>(src + off);
			else {
				uint8_t *vb = reinterpret_cast<uint8_t *>(&v);
				for (int k = 0; k < 16; k++)
					vb[k] = off + k < iend ? src[off + k] : 0;
			}
			*reinterpret_cast<uint4 *>(ring + (off & (RING - 1))) = v;
		}
		const uint32_t keep = target > RING ? target - RING : 0; // older bytes than this were overwritten
		if (from != W.hi)
			W.lo = from; // jumped ahead: only what was just loaded is valid
		else if (W.lo < keep)
			W.lo = keep;
		W.hi = target;
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
	};
	// table accessors
	auto tget = [&](uint32_t h) -> uint32_t { return by_u16 ? (uint32_t)table16[h] : table32[h]; };
	auto tset = [&](uint32_t h, uint32_t v) {
		if (by_u16)
			table16[h] = (uint16_t)v;
		else
			table32[h] = v;
	};

	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
	ensure(0);

	// wave-uniform automaton state (positions are offsets from src)
	uint32_t ip = 0, anchor = 0, match = 0;
	long long op = 0;
	int result = -1;

	if (src_size < LZ4_MINLENGTH)
		goto last_literals;

	if (lane == 0)
		tset(W.hash(0, by_u16), 0);
	ip = 1;

	for (;;) {
		// ---- find a match: 64 probes of the greedy scan per round, one per lane ----
		// The skip acceleration is a pure function of the probe count, so the position of every
		// probe of a search is known in closed form.  Every lane hashes its position and
		// reads the table as it was before the round; a probe that an EARLIER probe of the same
		// round would have overwritten (same hash) takes that probe's position instead, so each
		// lane sees exactly the table state of the serial automaton.  The first lane that finds a
		// match (or runs into the end of the input) wins; only the probes up to it insert.
		// probe t of a search (t = 0, 1, ...) advances by step_t = 1 for t = 0, (63 + t) >> 6 after
		// (liblz4: step = searchMatchNb++ >> skipTrigger, applied one probe late), so it looks at
		// start + 1 + F(63 + t) with F(x) = sum_{c<x} (c >> 6) = 32 a (a - 1) + a b, a = x >> 6, b = x & 63
		const uint32_t start = ip;
		auto pos_of = [&](uint32_t t) -> uint64_t {
			if (t == 0)
				return start;
			const uint64_t x = 63ull + t, a = x >> LZ4_SKIPTRIGGER, b = x & 63;
			return (uint64_t)start + 1 + 32 * a * (a - 1) + a * b;
		};
		uint32_t T = 0; // probes done in this search
		for (;;) {
			const uint32_t t = T + (uint32_t)lane;
			const uint64_t pos64 = pos_of(t);
			const uint64_t nxt64 = pos64 + (t == 0 ? 1 : (63ull + t) >> LZ4_SKIPTRIGGER);
			const bool endk = nxt64 > mflimit_plus_one; // includes pos > mflimit_plus_one
			const uint32_t pos = (uint32_t)pos64;
			ensure((uint32_t)__shfl(pos, 0));
			uint32_t h = 0, cur4 = 0, m = 0;
			if (!endk) {
				if (by_u16) {
					cur4 = W.rd32(pos);
					h = (cur4 * 2654435761U) >> (MINMATCH * 8 - (LZ4_HASHLOG + 1));
				} else {
					const uint64_t v = W.rd64(pos);
					cur4 = (uint32_t)v;
					h = (uint32_t)(((v << 24) * 889523592379ULL) >> (64 - LZ4_HASHLOG));
				}
				m = tget(h);
				owner[h] = (uint8_t)lane;
			}
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
			__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
			// probes of this round that share a hash: later ones see the earlier one's position
			unsigned long long grp = 0;
			unsigned long long pending = __ballot(!endk && owner[h] != (uint8_t)lane);
			while (pending) {
				const int j = __ffsll((long long)pending) - 1;
				const uint32_t hj = __shfl(h, j);
				const bool mine = !endk && h == hj;
				const unsigned long long G = __ballot(mine);
				if (mine) {
					grp = G;
					const unsigned long long below = G & ((1ull << lane) - 1);
					if (below) {
						const uint32_t jj = 63 - __clzll((long long)below);
						m = (uint32_t)pos_of(T + jj);
					}
				}
				pending &= ~G;
			}
			const bool found = !endk && (by_u16 || m + LZ4_DISTANCE_MAX >= pos) && W.rd32(m) == cur4;
			const unsigned long long stop = __ballot(found || endk);
			if (!stop) {
				if ((grp >> lane) >> 1 == 0)
					tset(h, pos);
				T += 64;
				__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
				__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
				continue;
			}
			const int f = __ffsll((long long)stop) - 1;
			const bool is_end = __shfl((int)endk, f) != 0;
			// probes 0..f insert (the probe that hit the end does not)
			const int upto = is_end ? f - 1 : f;
			if (lane <= upto) {
				const unsigned long long later = grp & ~((2ull << lane) - 1) & (upto >= 63 ? ~0ull : (1ull << (upto + 1)) - 1);
				if (!later)
					tset(h, pos);
			}
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
			__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
			if (is_end)
				goto last_literals;
			ip = __shfl(pos, f);
			match = __shfl(m, f);
			break;
		}
		// catch up: extend the match backwards, 64 bytes per step
		{
			const uint32_t maxback = ip - anchor < match ? ip - anchor : match;
			uint32_t back = 0;
			while (back < maxback) {
				const uint32_t k = back + (uint32_t)lane;
				const bool ok = k < maxback && W.rd8(ip - 1 - k) == W.rd8(match - 1 - k);
				const unsigned long long bad = __ballot(!ok);
				if (bad) {
					back += (uint32_t)(__ffsll((long long)bad) - 1);
					break;
				}
				back += 64;
			}
			ip -= back;
			match -= back;
		}
		// ---- literals ----
		{
			unsigned lit = ip - anchor;
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
			unsigned mc = W.count_eq(ip + MINMATCH, match + MINMATCH, matchlimit, lane);
			ip += mc + MINMATCH;
			if (limited && op + (1 + LASTLITERALS) + (mc + 240) / 255 > olimit) {
				result = 0;
				goto done;
			}
			if (mc >= ML_MASK)
				op += (mc - ML_MASK) / 255 + 1;
		}
		anchor = ip;
		if (job.stop_below > 0) {
			const long long rest = (long long)iend - (long long)anchor;
			const long long ub = op + rest + rest / 255 + 16;
			if (ub < (long long)job.stop_below) {
				result = (int)ub;
				goto done;
			}
		}
		if (ip >= mflimit_plus_one)
			break;
		ensure(ip);
		{
			int again = 0;
			uint32_t r_midx = 0;
			if (lane == 0) {
				tset(W.hash(ip - 2, by_u16), ip - 2);
				const uint32_t h = W.hash(ip, by_u16);
				const uint32_t match_index = tget(h);
				tset(h, ip);
				r_midx = match_index;
				if ((by_u16 || match_index + LZ4_DISTANCE_MAX >= ip) && W.rd32(match_index) == W.rd32(ip))
					again = 1;
			}
			again = __shfl(again, 0);
			r_midx = __shfl(r_midx, 0);
			if (again) {
				match = r_midx;
				op++; // token, zero literals
				goto next_match;
			}
		}
		ip++;
	}

last_literals:
	{
		long long last_run = (long long)iend - (long long)anchor;
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
	const size_t lds = (size_t)RING + TABLE_BYTES + OWNER_BYTES;
	static int attr_rc = (int)hipFuncSetAttribute((const void *)k_lz4_size, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
	if (attr_rc != (int)hipSuccess)
		return -1;
	hipLaunchKernelGGL(k_lz4_size, dim3(njobs), dim3(64), lds, s, d_jobs, d_results);
	return hipGetLastError() == hipSuccess ? 0 : -1;
}

} // namespace lrzgpu
