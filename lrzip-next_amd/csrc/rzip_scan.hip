// rzip_scan.hip -- the rzip long-range-match preprocessor on MI355X (gfx950).
//
// Reference: src/rzip.c hash_search (586-762) with insert_hash (304-353), clean_one_from_hash
// (357-383), find_best_match (495-534), single_match_len (431-461), tags (385-416).
// The reference is one CPU loop over every byte.  Here the work is split by what is and is not
// order-dependent:
//
//   K1 k_tag_scan      all CUs.  Rolling 31-byte XOR tags for every position of a segment from an
//                      LDS-staged byte tile (coalesced 16 B HBM reads), filtered by the resolver's
//                      CURRENT minimum tag mask (masks only ever gain bits, so the filter yields a
//                      superset of every later lookup/insert), compacted in position order with a
//                      workgroup prefix sum.  HBM-bound: 1 B read per position.
//      k_tile_scan /   exclusive scan of the per-tile counts and the packed, position-ordered
//      k_compact_cands candidate list of the segment (64 candidates per resolver load at any density).
//   K2 k_resolve_mw<4> one workgroup of four wavefronts (rzip_resolve_mw.h; the one-wavefront kernel it
//                      grew out of: tools/experiments/resolver_one_wavefront.hip.inc).  The exact
//                      hash-table automaton over the candidates, as a
//                      speculative in-order window: 256 candidates, one per lane, are simulated
//                      against the table as it stands (walks over one rank byte and one fingerprint
//                      byte per slot, 64 slots per step, 8 slots per SWAR operation; twins with the
//                      same tag on top of their predecessor's predicted insert), the longest prefix
//                      that is provably what the serial automaton would do is committed in parallel
//                      (hashed LDS write counters + exact check for conflicts, ballot prefix counts
//                      for the clean sweep), everything else -- real matches, sweep wraps, lazy
//                      matching, collapsed tag spaces -- takes the serial step: one coalesced 64-slot
//                      window load + wave ballots per probe, LDS stack for insert displacement,
//                      64-lane wide match verification.  Bound by instruction issue and dependent
//                      loads of one wave (serial table state), bit-exact by construction.
//   K3 k_long_compare  all CUs.  Forward extent of a match that is still equal after 4 MiB; the
//                      resolver launch ends with the request and resumes with the answer as a hint.
//   K4 k_gather_runs   all CUs.  Materialises stream 1 (literal bytes) from the run table.
//   K5 k_crc32_tiles   all CUs.  CRC-32 of the chunk, 64 KiB tiles combined with GF(2) shifts.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>

#include <chrono>

#include "profile.h"
#include "common.h"
#include "rzip_scan.h"
#include "rzip_census.h"

namespace lrzgpu {

#define HIPCHK(x)                                                                              \
	do {                                                                                   \
		hipError_t e_ = (x);                                                           \
		if (e_ != hipSuccess) {                                                        \
			fprintf(stderr, "lrzgpu: HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
			return -1;                                                             \
		}                                                                              \
	} while (0)

constexpr int MINIMUM_MATCH = 31; // src/rzip.c:51
constexpr int GREAT_MATCH = 1024; // src/rzip.c:50
constexpr int TILE = 4096;        // positions per K1 workgroup
constexpr int PER_THREAD = 16;
constexpr long long LONG_EXTENT = 4 << 20; // forward extents longer than this go to k_long_compare
constexpr int A1_MAX_STEPS = 8192; // slots a speculative walk may cover (longer clusters: serial path)
constexpr int MAX_EQS = 32;  // round-robin eviction handled in the batch up to this max_chain_len
constexpr int MAX_HITS = 24; // tag hits one speculative lookup may verify (more -> serial path)
constexpr int DENSE_HITS = 40; // ... in the dense variant (one wavefront: its hits and their measured extents have the LDS to themselves)
// the 8-wavefront resolver trades both for window: 160 KB of LDS hold 512 candidates with these (levels <= 7 only)
constexpr int CF_BITS = 13; // conflict filter: 16-bit write counters per hashed 8-slot granule (<= 320 writes per round)

typedef unsigned long long u64;
typedef long long i64;

struct __attribute__((aligned(16))) Slot {
	i64 offset;
	u64 t;
};

struct __attribute__((packed, aligned(1))) U64u {
	u64 v;
};
struct __attribute__((packed, aligned(1))) U128u {
	u64 a, b;
};

void rzip_level_params(int level, unsigned *mb_used, unsigned *initial_freq, unsigned *max_chain_len)
{
	// src/rzip.c:67-82
	static const unsigned L[10][3] = {{1, 4, 1}, {2, 4, 2}, {4, 4, 2}, {8, 4, 2}, {16, 4, 3},
					  {32, 4, 4}, {32, 2, 6}, {64, 1, 16}, {64, 1, 32}, {64, 1, 128}};
	if (level < 0) level = 0;
	if (level > 9) level = 9;
	*mb_used = L[level][0];
	*initial_freq = L[level][1];
	*max_chain_len = L[level][2];
}

void hash_index_table(uint64_t out[256])
{
	// src/rzip.c:765-771: hash_index[i] = (random() << 16) ^ random() with glibc's default-seeded
	// TYPE_3 additive feedback generator (the reference never seeds it). Frozen here so the table
	// does not depend on libc or on other random() users in the process.
	static int32_t r[34 + 310 + 512];
	r[0] = 1;
	for (int i = 1; i < 31; i++) {
		long hi = r[i - 1] / 127773, lo = r[i - 1] % 127773;
		long w = 16807 * lo - 2836 * hi;
		if (w < 0)
			w += 2147483647;
		r[i] = (int32_t)w;
	}
	for (int i = 31; i < 34; i++)
		r[i] = r[i - 31];
	for (int i = 34; i < 34 + 310 + 512; i++)
		r[i] = (int32_t)((uint32_t)r[i - 31] + (uint32_t)r[i - 3]);
	int k = 0;
	for (int i = 0; i < 256; i++) {
		uint64_t a = ((uint32_t)r[344 + k++]) >> 1;
		uint64_t b = ((uint32_t)r[344 + k++]) >> 1;
		out[i] = (a << 16) ^ b;
	}
}

// ---------------------------------------------------------------------------------------------
// K1: tags + candidate compaction
// ---------------------------------------------------------------------------------------------
// Round 6.  tag(p) = XOR of hash_index[b] over the 31 bytes from p on (src/rzip.c:385-416) = X(p + 31) ^ X(p) with
// X(i) = the XOR over all bytes in front of i: ONE table lookup per position, a prefix XOR over the tile (16 positions per
// thread in registers, the thread totals through a wavefront scan, the wave totals through LDS), and X(p + 31) from the
// thread one or two places on through a transposed LDS array (lane t reads lane t + 1's or t + 2's column: consecutive
// lanes, consecutive addresses).  Rounds 1 to 5 rolled every thread's tag by itself: 61 table reads and 61 byte reads per
// 16 positions, the byte reads 16 bytes apart from lane to lane -- all of them on eight banks.  The tile's bytes come
// straight from memory, one aligned 16-byte word per thread: the host starts a segment on a multiple of 16 (positions in
// front of the first candidate that counts are dropped by the resolver's `pos > p_skip` as before).
__global__ void __launch_bounds__(256) k_tag_scan(const uint8_t *__restrict__ buf, i64 seg_lo, i64 seg_hi,
						  const u64 *__restrict__ hx_g, const ScanState *__restrict__ st,
						  uint32_t *__restrict__ cand_rel, u64 *__restrict__ cand_tag,
						  uint32_t *__restrict__ tile_count)
{
	__shared__ u64 hx[256];
	// xs[j & 7][t] = X(16 t + j), in two halves (j < 8, then j >= 8: 16.5 KB instead of 33 -- with 35 KB a workgroup did not
	// fit beside the finder's walk workgroups on a CU and waited for one to drain: 8 ms a launch inside the pipeline
	// against 0.3 ms alone); columns 256, 257: the 32 positions behind the tile
	__shared__ u64 xs[PER_THREAD / 2][256 + 2];
	__shared__ u64 wave_x[4];
	__shared__ uint32_t wave_tot[4];

	const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
	const i64 p0 = seg_lo + (i64)blockIdx.x * TILE; // (a multiple of 16, like the chunk's base: caller contract)
	const u64 min_mask = st->min_mask;
	hx[tid] = hx_g[tid];
	i64 need = seg_hi - p0;
	if (need > TILE)
		need = TILE;
	const i64 nbytes = need + (MINIMUM_MATCH - 1); // (the chunk's allocation is readable 64 B past its end)
	uint4 w = make_uint4(0, 0, 0, 0), h0 = w, h1 = w;
	if ((i64)tid * 16 < nbytes)
		w = *reinterpret_cast<const uint4 *>(buf + p0 + tid * 16);
	if (tid < 2 && TILE < nbytes)
		h0 = *reinterpret_cast<const uint4 *>(buf + p0 + TILE);
	if (tid == 1 && TILE + 16 < nbytes)
		h1 = *reinterpret_cast<const uint4 *>(buf + p0 + TILE + 16);
	__syncthreads();

	auto byte_of = [](const uint4 &q, int j) -> uint32_t {
		const uint32_t d = j < 4 ? q.x : j < 8 ? q.y : j < 12 ? q.z : q.w;
		return (d >> (8 * (j & 3))) & 0xFFu;
	};
	u64 x[PER_THREAD]; // inclusive prefix XOR over my 16 bytes
	u64 acc = 0;
#pragma unroll
	for (int j = 0; j < PER_THREAD; j++) {
		acc ^= hx[byte_of(w, j)];
		x[j] = acc;
	}
	u64 inc = acc; // inclusive scan of the thread totals over the wavefront
#pragma unroll
	for (int d = 1; d < 64; d <<= 1) {
		const u64 o = ((u64)__shfl_up((uint32_t)(inc >> 32), d) << 32) | __shfl_up((uint32_t)inc, d);
		if (lane >= d)
			inc ^= o;
	}
	if (lane == 63)
		wave_x[wv] = inc;
	__syncthreads();
	u64 excl = inc ^ acc; // everything in front of my first byte
	for (int q = 0; q < wv; q++)
		excl ^= wave_x[q];
	u64 xh[PER_THREAD]; // the 32 positions behind the tile (threads 0 and 1)
	if (tid < 2) {
		u64 a = wave_x[0] ^ wave_x[1] ^ wave_x[2] ^ wave_x[3]; // X(4096)
		if (tid == 1) {
#pragma unroll
			for (int j = 0; j < 16; j++)
				a ^= hx[byte_of(h0, j)];
		}
		const uint4 mine = tid ? h1 : h0;
#pragma unroll
		for (int j = 0; j < PER_THREAD; j++) {
			xh[j] = a;
			a ^= hx[byte_of(mine, j)];
		}
	}
	// tags of my 16 positions: X(i + 31) ^ X(i), i = 16 tid + k; X(i + 31) = X(16 (tid + 1) + 15) for k = 0, else
	// X(16 (tid + 2) + k - 1): the half with j < 8 serves k = 1 .. 8, the half with j >= 8 serves k = 9 .. 15 and k = 0
	const int l0 = tid * PER_THREAD;
	u64 tags[PER_THREAD];
	uint32_t bits = 0;
#pragma unroll
	for (int half = 0; half < 2; half++) {
		if (half)
			__syncthreads(); // (the first half has been read)
#pragma unroll
		for (int j = 0; j < PER_THREAD / 2; j++) {
			const int jj = 8 * half + j;
			xs[j][tid] = jj ? excl ^ x[jj - 1] : excl;
			if (tid < 2)
				xs[j][256 + tid] = xh[jj];
		}
		__syncthreads();
#pragma unroll
		for (int k = 0; k < PER_THREAD; k++) {
			const int jn = (k + 15) & 15; // X(i + 31) is entry jn of column tid + 1 + ((k + 15) >> 4)
			if ((jn >> 3) != half)
				continue;
			const u64 xi = k ? excl ^ x[k - 1] : excl;
			const u64 t = xs[jn & 7][tid + 1 + ((k + 15) >> 4)] ^ xi;
			tags[k] = t;
			if (l0 + k < need && (t & min_mask) == min_mask)
				bits |= 1u << k;
		}
	}
	// workgroup exclusive scan of popcounts
	uint32_t cnt = __popc(bits);
	uint32_t incl = cnt;
#pragma unroll
	for (int d = 1; d < 64; d <<= 1) {
		uint32_t o = __shfl_up(incl, d);
		if (lane >= d)
			incl += o;
	}
	if (lane == 63)
		wave_tot[wv] = incl;
	__syncthreads();
	uint32_t base = 0;
	for (int q = 0; q < wv; q++)
		base += wave_tot[q];
	uint32_t pos = base + incl - cnt;
	const size_t out0 = (size_t)blockIdx.x * TILE;
#pragma unroll
	for (int k = 0; k < PER_THREAD; k++)
		if (bits & (1u << k)) {
			cand_rel[out0 + pos] = (uint32_t)((i64)blockIdx.x * TILE + l0 + k);
			cand_tag[out0 + pos] = tags[k];
			pos++;
		}
	if (tid == 255)
		tile_count[blockIdx.x] = base + incl;
}

// K1b: exclusive scan of the per-tile candidate counts (one workgroup; a segment has <= 65536 tiles).
// tile_base[t] = candidates before tile t, tile_base[ntiles] = all of them.
__global__ void __launch_bounds__(1024) k_tile_scan(const uint32_t *__restrict__ tile_count, int ntiles, uint32_t *__restrict__ tile_base)
{
	__shared__ uint32_t wsum[16];
	const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
	const int per = (ntiles + 1023) / 1024;
	const int t0 = tid * per;
	uint32_t mine = 0;
	for (int k = 0; k < per; k++)
		if (t0 + k < ntiles)
			mine += tile_count[t0 + k];
	uint32_t incl = mine;
#pragma unroll
	for (int d = 1; d < 64; d <<= 1) {
		const uint32_t o = __shfl_up(incl, d);
		if (lane >= d)
			incl += o;
	}
	if (lane == 63)
		wsum[wv] = incl;
	__syncthreads();
	uint32_t base = 0;
	for (int w = 0; w < wv; w++)
		base += wsum[w];
	uint32_t run = base + incl - mine;
	for (int k = 0; k < per; k++)
		if (t0 + k < ntiles) {
			tile_base[t0 + k] = run;
			run += tile_count[t0 + k];
		}
	if (tid == 1023)
		tile_base[ntiles] = base + incl;
}

// K1c: the per-tile candidate lists, packed into one list in position order (the resolver then
// reads 64 candidates per load whatever their density; sparse masks leave ~8 per tile).
__global__ void __launch_bounds__(256) k_compact_cands(const uint32_t *__restrict__ cand_rel, const u64 *__restrict__ cand_tag,
						       const uint32_t *__restrict__ tile_count, const uint32_t *__restrict__ tile_base,
						       uint32_t cap, uint32_t *__restrict__ comp_rel, u64 *__restrict__ comp_tag)
{
	const uint32_t cnt = tile_count[blockIdx.x], base = tile_base[blockIdx.x];
	const size_t in0 = (size_t)blockIdx.x * TILE;
	for (uint32_t i = threadIdx.x; i < cnt; i += 256)
		if (base + i < cap) {
			comp_rel[base + i] = cand_rel[in0 + i];
			comp_tag[base + i] = cand_tag[in0 + i];
		}
}

// ---------------------------------------------------------------------------------------------
// K2: the resolver (one wavefront)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ u64 bcast64(u64 v, int src)
{
	uint32_t lo = __shfl((uint32_t)v, src), hi = __shfl((uint32_t)(v >> 32), src);
	return ((u64)hi << 32) | lo;
}
// the same when every lane asks for the same source lane (src is wave-uniform): no trip through the LDS crossbar
__device__ __forceinline__ u64 readlane64(u64 v, int src)
{
	const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, src), hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), src);
	return ((u64)hi << 32) | lo;
}
__device__ __forceinline__ u64 mask_up(u64 m) { return (m << 1) | 1; }
__device__ __forceinline__ int bitness_rank(u64 t) // ffsll(~t)
{
	u64 v = ~t;
	return v ? __ffsll((long long)v) : 0;
}
__device__ __forceinline__ u64 low_mask(int n) { return n >= 64 ? ~0ull : ((1ull << n) - 1); }

// ---- side arrays of the hash table: one rank byte and one fingerprint byte per slot -------------
// The walks of hash_search only ask three things of a slot: is it empty, does it stop an insert
// (empty, due for cleaning, or of lesser bitness than the tag being inserted), and does it hold the
// same tag.  With rank = min(ffsll(~t), 63) for an occupied slot and 0 for an empty one, the first
// two are `rank == 0` and `rank < max(my_rank, popcount(better) + 1)`; the third is pre-filtered by a
// fingerprint (tag bits 32..39) and confirmed on the 16-byte slot.  A walk then reads 2 bytes per
// slot instead of 16 and tests 8 slots per 64-bit SWAR step.
constexpr u64 B80 = 0x8080808080808080ull, B7F = 0x7F7F7F7F7F7F7F7Full, B01 = 0x0101010101010101ull;
__device__ __forceinline__ uint8_t rank_byte(u64 t, i64 off)
{
	if (!(t | (u64)off))
		return 0;
	const int r = bitness_rank(t);
	return (uint8_t)(r < 63 ? r : 63);
}
__device__ __forceinline__ uint8_t fp_byte(u64 t) { return (uint8_t)(t >> 32); }
// 0x80 in every byte of x that is < the byte replicated in thr8 (all bytes < 128)
__device__ __forceinline__ u64 bytes_lt(u64 x, u64 thr8) { return ~((x | B80) - thr8) & B80; }
// 0x80 in every byte of x that equals the byte replicated in c8
__device__ __forceinline__ u64 bytes_eq(u64 x, u64 c8)
{
	const u64 y = x ^ c8;
	return ~(((y & B7F) + B7F) | y | B7F);
}
// the eight 0x80 flags of f as eight bits (flag of byte i -> bit i)
__device__ __forceinline__ uint32_t flags_to_bits(u64 f)
{
	const uint32_t lo = (uint32_t)f >> 7, hi = (uint32_t)(f >> 32) >> 7;
	return (((lo * 0x01020408u) >> 24) & 0xFu) | (((hi * 0x01020408u) >> 20) & 0xF0u);
}
// index of the k-th (0-based) set bit of m; m must have more than k bits set
__device__ __forceinline__ int nth_set_bit(u64 m, int k)
{
	for (int i = 0; i < k; i++)
		m &= m - 1;
	return __ffsll((long long)m) - 1;
}

// Cheap exact pre-test for a tag hit: true when single_match_len(p0, op) is certainly 0, decided
// from the 8 bytes after and the 8 bytes before the two positions (a match needs fwd + rev >= 31;
// two early mismatches bound the total by 14).  false = undecided, run the exact compare.
__device__ __forceinline__ bool quick_reject(const uint8_t *buf, i64 p0, i64 op, i64 end, i64 last_match)
{
	if (op >= p0)
		return true;
	if (end - p0 < 8)
		return false;
	const u64 fa = reinterpret_cast<const U64u *>(buf + p0)->v;
	const u64 fb = reinterpret_cast<const U64u *>(buf + op)->v;
	if (fa == fb)
		return false;
	const i64 floor_p = last_match > 0 ? last_match : 0;
	i64 max_back = p0 - floor_p;
	if (op < max_back)
		max_back = op;
	if (max_back < 8)
		return true; // fwd < 8 and rev <= max_back < 8
	const u64 ba = reinterpret_cast<const U64u *>(buf + p0 - 8)->v;
	const u64 bb = reinterpret_cast<const U64u *>(buf + op - 8)->v;
	return ba != bb; // some byte among the 8 before differs: rev < 8
}

// single_match_len(p0, op) computed by ONE lane, when both extents end inside their bounds (most tag hits that are real
// matches at all are short ones: a phrase, a line); -1 = an extent reached its bound, take the wave-wide compare.
// The serial lookup verifies all the hits of a probe window with this in one round of memory latency instead of one
// wave-wide compare (two to three dependent round trips) per hit.
__device__ __forceinline__ i64 lane_match_len(const uint8_t *buf, i64 p0, i64 op, i64 end, i64 last_match, i64 *rev)
{
	constexpr i64 FWD_BOUND = 256, BACK_BOUND = 128;
	*rev = 0;
	if (op >= p0)
		return 0;
	i64 total = end - p0;
	if (total < 0)
		total = 0;
	const i64 capf = total < FWD_BOUND ? total : FWD_BOUND;
	i64 fwd = 0;
	while (fwd + 8 <= capf) {
		const u64 x = reinterpret_cast<const U64u *>(buf + p0 + fwd)->v ^ reinterpret_cast<const U64u *>(buf + op + fwd)->v;
		if (x) {
			fwd += (__ffsll((long long)x) - 1) >> 3;
			goto fwd_done;
		}
		fwd += 8;
	}
	while (fwd < capf && buf[p0 + fwd] == buf[op + fwd])
		fwd++;
	if (fwd == capf && capf < total)
		return -1;
fwd_done:;
	const i64 floor_p = last_match > 0 ? last_match : 0;
	i64 max_back = p0 - floor_p;
	if (op < max_back)
		max_back = op;
	if (max_back < 0)
		max_back = 0;
	const i64 capb = max_back < BACK_BOUND ? max_back : BACK_BOUND;
	i64 back = 0;
	while (back + 8 <= capb) {
		const u64 x = reinterpret_cast<const U64u *>(buf + p0 - 8 - back)->v ^ reinterpret_cast<const U64u *>(buf + op - 8 - back)->v;
		if (x) {
			back += (i64)(__clzll((long long)x) >> 3); // equal bytes from the top: the ones next to p0
			goto back_done;
		}
		back += 8;
	}
	while (back < capb && buf[p0 - 1 - back] == buf[op - 1 - back])
		back++;
	if (back == capb && capb < max_back)
		return -1;
back_done:;
	*rev = back;
	const i64 len = fwd + back;
	return len < MINIMUM_MATCH ? 0 : len;
}


struct Resolver {
	const uint8_t *buf;
	Slot *tbl;
	uint8_t *rk, *fpa; // rank / fingerprint byte per slot
	u64 hmask;
	i64 end, last_match;
	u64 tag_mask, min_mask;
	i64 hash_count, hash_limit, clean_ptr, victim_round;
	uint32_t max_chain;
	i64 tag_hits, tag_misses;
	int lane;
	// LDS stack for displaced entries
	u64 *stk_t;
	i64 *stk_off, *stk_h;
	// long forward extents are computed by the whole GPU (k_long_compare): hint = a finished one,
	// ext_* = the request this launch ends with
	i64 hint_p, hint_op, hint_len;
	bool allow_abort;
	mutable bool aborted;
	mutable i64 ext_p, ext_op, ext_done;

	// every table write goes through here: the slot and its two side bytes
	__device__ __forceinline__ void store_slot(i64 slot, u64 t, i64 off) const
	{
		Slot w;
		w.offset = off;
		w.t = t;
		tbl[slot] = w;
		rk[slot] = rank_byte(t, off);
		fpa[slot] = fp_byte(t);
	}

	// Forward extent: number of equal bytes of buf[p..] and buf[op..], p bounded by `end`.
	__device__ i64 extent_fwd(i64 p, i64 op) const
	{
		const i64 total = end - p;
		if (total <= 0)
			return 0;
		i64 done = 0;
		// phase 1: 512 B per step (8 B per lane) -- false tag positives die here
		for (int it = 0; it < 4; it++) {
			i64 off = done + (i64)lane * 8;
			int mism = 8;
			if (off < total) {
				i64 lim = total - off < 8 ? total - off : 8;
				if (lim == 8) {
					u64 a = reinterpret_cast<const U64u *>(buf + p + off)->v;
					u64 b = reinterpret_cast<const U64u *>(buf + op + off)->v;
					u64 x = a ^ b;
					mism = x ? (__ffsll((long long)x) - 1) >> 3 : 8;
				} else {
					mism = (int)lim;
					for (int k = 0; k < (int)lim; k++)
						if (buf[p + off + k] != buf[op + off + k]) {
							mism = k;
							break;
						}
				}
			} else
				mism = 0;
			u64 stop = __ballot(mism < 8);
			if (stop) {
				int first = __ffsll((long long)stop) - 1;
				i64 r = done + (i64)first * 8 + __builtin_amdgcn_readlane(mism, first);
				return r < total ? r : total;
			}
			done += 512;
			if (done >= total)
				return total;
		}
		// phase 2a: whole 16 KiB steps while everything is equal: 32 unconditional 16-byte loads per
		// lane in flight, one ballot per step (a multi-GiB copy streams at wave bandwidth)
		if (p == hint_p && op == hint_op)
			return hint_len < total ? hint_len : total;
		while (done + 16384 <= total) {
			if (done >= LONG_EXTENT && allow_abort && total - done >= LONG_EXTENT) {
				aborted = true; // the caller unwinds; the host runs the wide compare and relaunches
				ext_p = p;
				ext_op = op;
				ext_done = done;
				return done;
			}
			const uint8_t *pa = buf + p + done + (i64)lane * 16;
			const uint8_t *pb = buf + op + done + (i64)lane * 16;
			U128u va[16], vb[16];
#pragma unroll
			for (int c = 0; c < 16; c++) {
				va[c] = *reinterpret_cast<const U128u *>(pa + c * 1024);
				vb[c] = *reinterpret_cast<const U128u *>(pb + c * 1024);
			}
			u64 diff = 0;
#pragma unroll
			for (int c = 0; c < 16; c++)
				diff |= (va[c].a ^ vb[c].a) | (va[c].b ^ vb[c].b);
			if (__ballot(diff != 0))
				break; // locate the byte with the exact step below
			done += 16384;
		}
		if (done >= total)
			return total;
		// phase 2b: 16 KiB per step with exact mismatch location
		for (;;) {
			int mism_chunk = 16; // first differing 16-byte piece among my 16
			int mism_byte = 0;
			const i64 base = done + (i64)lane * 16;
#pragma unroll
			for (int c = 15; c >= 0; c--) {
				i64 off = base + (i64)c * 1024;
				if (off + 16 <= total) {
					U128u a = *reinterpret_cast<const U128u *>(buf + p + off);
					U128u b = *reinterpret_cast<const U128u *>(buf + op + off);
					u64 x0 = a.a ^ b.a, x1 = a.b ^ b.b;
					if (x0 | x1) {
						mism_chunk = c;
						mism_byte = x0 ? (__ffsll((long long)x0) - 1) >> 3 : 8 + ((__ffsll((long long)x1) - 1) >> 3);
					}
				} else if (off < total) {
					int lim = (int)(total - off), m = lim;
					for (int k = 0; k < lim; k++)
						if (buf[p + off + k] != buf[op + off + k]) {
							m = k;
							break;
						}
					mism_chunk = c;
					mism_byte = m;
				} else {
					mism_chunk = c;
					mism_byte = 0;
				}
			}
			// earliest mismatch position in this 16 KiB step, across lanes
			i64 mypos = mism_chunk < 16 ? (i64)mism_chunk * 1024 + (i64)lane * 16 + mism_byte : (i64)1 << 40;
			i64 best = mypos;
#pragma unroll
			for (int d = 32; d >= 1; d >>= 1) {
				i64 o = (i64)bcast64((u64)best, (lane ^ d));
				if (o < best)
					best = o;
			}
			best = (i64)readlane64((u64)best, 0);
			if (best < ((i64)1 << 40)) {
				i64 r = done + best;
				return r < total ? r : total;
			}
			done += 16384;
			if (done >= total)
				return total;
		}
	}

	// Backward extent: equal bytes buf[p-1-k] == buf[op-1-k], k < max_back.
	__device__ i64 extent_back(i64 p, i64 op, i64 max_back) const
	{
		if (max_back <= 0)
			return 0;
		i64 done = 0;
		for (;;) {
			i64 off = done + (i64)lane * 8; // bytes p-1-off .. p-8-off
			int mism = 8;
			if (off < max_back) {
				i64 lim = max_back - off < 8 ? max_back - off : 8;
				mism = (int)lim;
				for (int k = 0; k < (int)lim; k++)
					if (buf[p - 1 - off - k] != buf[op - 1 - off - k]) {
						mism = k;
						break;
					}
				if (lim == 8 && mism == 8)
					mism = 8;
			} else
				mism = 0;
			u64 stop = __ballot(mism < 8);
			if (stop) {
				int first = __ffsll((long long)stop) - 1;
				i64 r = done + (i64)first * 8 + __builtin_amdgcn_readlane(mism, first);
				return r < max_back ? r : max_back;
			}
			done += 512;
			if (done >= max_back)
				return max_back;
		}
	}

	// single_match_len, src/rzip.c:431-461
	__device__ i64 match_len(i64 p0, i64 op, i64 *rev) const
	{
		if (op >= p0)
			return 0;
		i64 len = extent_fwd(p0, op);
		i64 floor_p = last_match > 0 ? last_match : 0;
		i64 max_back = p0 - floor_p;
		if (op < max_back)
			max_back = op;
		i64 r = extent_back(p0, op, max_back);
		*rev = r;
		len += r;
		return len < MINIMUM_MATCH ? 0 : len;
	}

	// find_best_match, src/rzip.c:495-534
	__device__ i64 lookup(u64 t, i64 p, i64 *offset, i64 *reverse)
	{
		i64 best = 0;
		u64 h = t & hmask;
		*reverse = 0;
		for (;;) {
			Slot s = tbl[(h + lane) & hmask];
			bool empty = !(s.offset | (i64)s.t);
			u64 em = __ballot(empty);
			int first_empty = em ? __ffsll((long long)em) - 1 : 64;
			const bool is_hit = !empty && s.t == t && lane < first_empty;
			const u64 all_hits = __ballot(is_hit);
			// every hit lane pre-tests its own slot in parallel (one memory round trip for all of
			// them); only the undecided ones take the exact wave-wide compare, in slot order
			const bool undecided = is_hit && !quick_reject(buf, p, s.offset, end, last_match);
			u64 hits = __ballot(undecided);
			tag_misses += __popcll(all_hits) - __popcll(hits);
			// every undecided lane measures its own candidate (bounded); the loop below only falls back to the
			// wave-wide compare for the ones that ran into a bound
			i64 l_rev = 0, l_len = -1;
			if (undecided)
				l_len = lane_match_len(buf, p, s.offset, end, last_match, &l_rev);
			while (hits) {
				int idx = __ffsll((long long)hits) - 1;
				hits &= hits - 1;
				i64 cand_off = (i64)readlane64((u64)s.offset, idx);
				i64 rev = (i64)readlane64((u64)l_rev, idx);
				i64 mlen = (i64)readlane64((u64)l_len, idx);
				if (mlen < 0)
					mlen = match_len(p, cand_off, &rev);
				if (mlen) {
					if (mlen > best) {
						best = mlen;
						*offset = cand_off - rev;
						*reverse = rev;
					}
					tag_hits++;
				} else
					tag_misses++;
			}
			if (first_empty < 64)
				break;
			h = (h + 64) & hmask;
		}
		return best;
	}

	// insert_hash, src/rzip.c:304-353 (recursion unrolled onto an LDS stack)
	__device__ void insert(u64 t, i64 offset)
	{
		int depth = 0;
		u64 cur_t = t;
		i64 cur_off = offset;
		const u64 better = mask_up(min_mask);
		for (;;) {
			u64 h = cur_t & hmask;
			i64 round = 0;
			i64 victim_h = 0;
			const int my_rank = bitness_rank(cur_t);
			i64 write_h = -1;
			bool displaced = false;
			u64 disp_t = 0;
			i64 disp_off = 0;
			for (;;) {
				const u64 slot_idx = (h + lane) & hmask;
				Slot s = tbl[slot_idx];
				bool empty = !(s.offset | (i64)s.t);
				bool below = !empty && (s.t & better) != better;
				bool lesser = !empty && !below && bitness_rank(s.t) < my_rank;
				bool eq = !empty && !below && !lesser && s.t == cur_t;
				u64 stopm = __ballot(empty || below || lesser);
				int s1 = stopm ? __ffsll((long long)stopm) - 1 : 64;
				u64 eqm = __ballot(eq) & low_mask(s1);
				int neq = __popcll(eqm);
				// victim bookkeeping: the eq slot whose running index equals victim_round
				i64 need_v = victim_round - round;
				// chain limit inside this window?
				i64 left = (i64)max_chain - round; // eq slots still allowed before the limit
				if (neq >= left) {
					// the left-th eq slot (1-based) triggers the limit
					if (need_v >= 0 && need_v < left)
						victim_h = (i64)((h + (u64)nth_set_bit(eqm, (int)need_v)) & hmask);
					write_h = victim_h;
					hash_count--;
					if (++victim_round == (i64)max_chain)
						victim_round = 0;
					break;
				}
				if (need_v >= 0 && need_v < neq)
					victim_h = (i64)((h + (u64)nth_set_bit(eqm, (int)need_v)) & hmask);
				round += neq;
				if (s1 < 64) {
					const u64 sh = (h + (u64)s1) & hmask;
					bool s_empty = (__ballot(empty) >> s1) & 1;
					bool s_below = (__ballot(below) >> s1) & 1;
					if (s_empty) {
						write_h = (i64)sh;
					} else if (s_below) {
						hash_count--;
						write_h = (i64)sh;
					} else { // lesser bitness: rehash the occupant, then take its place
						displaced = true;
						disp_t = readlane64(s.t, s1);
						disp_off = (i64)readlane64((u64)s.offset, s1);
						write_h = (i64)sh;
					}
					break;
				}
				h = (h + 64) & hmask;
			}
			if (displaced) {
				if (lane == 0) {
					stk_t[depth] = cur_t;
					stk_off[depth] = cur_off;
					stk_h[depth] = write_h;
				}
				depth++;
				cur_t = disp_t;
				cur_off = disp_off;
				continue;
			}
			if (lane == 0)
				store_slot(write_h, cur_t, cur_off);
			break;
		}
		// unwind: outer frames overwrite the displaced occupant's old slot
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
		while (depth > 0) {
			depth--;
			if (lane == 0)
				store_slot(stk_h[depth], stk_t[depth], stk_off[depth]);
		}
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
	}

	// clean_one_from_hash, src/rzip.c:357-383
	__device__ u64 clean_one()
	{
		const i64 size = (i64)hmask + 1;
		for (;;) {
			const u64 better = mask_up(min_mask);
			while (clean_ptr < size) {
				i64 idx = clean_ptr + lane;
				bool cand = false;
				if (idx < size) {
					Slot s = tbl[idx];
					bool empty = !(s.offset | (i64)s.t);
					cand = !empty && (s.t & better) != better;
				}
				u64 m = __ballot(cand);
				if (m) {
					int k = __ffsll((long long)m) - 1;
					if (lane == k)
						store_slot(idx, 0, 0);
					clean_ptr += k;
					hash_count--;
					__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
					__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
					return better;
				}
				clean_ptr += 64;
			}
			min_mask = better;
			clean_ptr = 0;
		}
	}
};

// The hash_search automaton as wave 0 of k_resolve_mw carries it: the table side (Resolver) and the match in the making.
struct Automaton {
	Resolver R;
	i64 p_skip, cur_p, cur_ofs, cur_len;
	i64 n_rec, rec_cap, inserts, lookups;
	i64 serial_n; // exact steps taken
	MatchRec *records;
	int error;

	// what follows the lookup and the insert of a step (src/rzip.c:697-731): the longer match is kept, and one that is
	// long enough or far enough behind is emitted; true = emitted and the same candidate stands again
	__device__ __forceinline__ bool keep_or_emit(i64 P, u64 T, i64 mlen, i64 offset, i64 reverse)
	{
		if (mlen > cur_len) {
			cur_p = P - reverse;
			cur_len = mlen;
			cur_ofs = offset;
		}
		if (!((cur_len >= GREAT_MATCH || P >= cur_p + MINIMUM_MATCH) && cur_len >= MINIMUM_MATCH))
			return false;
		if (n_rec >= rec_cap) {
			error = 1;
			return false;
		}
		if (R.lane == 0) {
			MatchRec r;
			r.p = cur_p;
			r.ofs = cur_ofs;
			r.len = cur_len;
			records[n_rec] = r;
		}
		n_rec++;
		R.last_match = cur_p + cur_len;
		p_skip = R.last_match;
		cur_p = R.last_match;
		cur_len = 0;
		return P > p_skip && P <= R.end && (T & R.min_mask) == R.min_mask;
	}

	// One exact step at candidate (P, T), every case: the body of hash_search's loop, src/rzip.c:656-736.
	__device__ __forceinline__ void step(i64 P, u64 T)
	{
		serial_n++;
		R.allow_abort = true; // (a long extent may go to the whole GPU only before the candidate's first insert)
		bool again;
		do {
			i64 offset = 0, reverse = 0;
			lookups++;
			const i64 hits0 = R.tag_hits, misses0 = R.tag_misses;
			const i64 mlen = R.lookup(T, P, &offset, &reverse);
			if (R.aborted) {
				lookups--;
				serial_n--;
				R.tag_hits = hits0;
				R.tag_misses = misses0;
				if (P - 1 > p_skip)
					p_skip = P - 1;
				error = 3;
				return;
			}
			R.allow_abort = false;
			if ((T & R.tag_mask) == R.tag_mask) {
				inserts++;
				R.hash_count++;
				R.insert(T, P);
				if (R.hash_count > R.hash_limit)
					R.tag_mask = R.clean_one();
			}
			again = keep_or_emit(P, T, mlen, offset, reverse);
		} while (again && !error);
	}
};

// Per-lane speculative simulation of one automaton step against the table as it stands at the
// start of a round (see k_resolve_mw).  Everything it would read lies in [lo, hi]; everything it
// would write is listed in w_*.
struct LaneSim {
	bool complex_;   // must take the serial path (victim round-robin, wrap, deep displacement, ...)
	bool match;      // a tag hit verifies as a real match (>= MINIMUM_MATCH): batch ends here
	bool ins;        // (T & tag_mask) == tag_mask
	bool victim;     // insert meets max_chain_len equal tags: round-robin eviction (w_slot[0] set at commit)
	int dec;         // insert replaces an entry that was due for cleaning: hash_count--
	int misses;      // false tag positives met by the lookup
	int nw;          // table writes of the insert (displacement chain), <= 4
	uint32_t w_slot[4];
	u64 w_t[4];
	i64 w_off[4];
	uint32_t lo, hi; // inclusive slot interval read
	// twin successor: simulated on top of the insert its predecessor (same tag, previous lane) is
	// expected to make: (tag, predecessor position) written to tw_slot, whose occupant was of kind tw_kind
	// (0 empty, 1 due for cleaning, 2 of lesser bitness: displaced by the predecessor)
	bool twin;
	bool tw_over; // the twin's own tag is due for cleaning (before the first clean of a chunk): it overwrites tw_slot
	int tw_kind;
	int k0; // kind of the first stop of this lane's own insert (-1 none, 3 round-robin eviction)
	uint32_t tw_slot;
	// the dense variant (k_resolve_mw<1, ..., true>): the lookup's tag hits stay in LDS with their measured extents
	int nh;   // how many (<= MAXH)
	bool pot; // some hit may be a match (forward + backward extent >= MINIMUM_MATCH before last_match clips the latter)
	bool soft1; // my insert replaces an entry of MY OWN tag that was due for cleaning (kind 1): like a round-robin eviction it
	            // leaves the slot's rank byte, fingerprint byte and tag word as they are
};

// is there a match of at least MINIMUM_MATCH bytes between p0 and op? (single_match_len() != 0)
__device__ __forceinline__ bool lane_verify(const uint8_t *buf, i64 p0, i64 op, i64 end, i64 last_match)
{
	if (op >= p0)
		return false;
	i64 fwd_max = end - p0;
	if (fwd_max < 0)
		fwd_max = 0;
	i64 fwd = 0;
	const i64 cap = fwd_max < 32 ? fwd_max : 32;
	while (fwd + 8 <= cap) {
		u64 a = reinterpret_cast<const U64u *>(buf + p0 + fwd)->v;
		u64 b = reinterpret_cast<const U64u *>(buf + op + fwd)->v;
		u64 x = a ^ b;
		if (x) {
			fwd += (__ffsll((long long)x) - 1) >> 3;
			goto fwd_done;
		}
		fwd += 8;
	}
	while (fwd < cap && buf[p0 + fwd] == buf[op + fwd])
		fwd++;
fwd_done:
	if (fwd >= MINIMUM_MATCH)
		return true;
	i64 floor_p = last_match > 0 ? last_match : 0;
	i64 max_back = p0 - floor_p;
	if (op < max_back)
		max_back = op;
	const i64 need = MINIMUM_MATCH - fwd;
	if (max_back < need)
		return false;
	for (i64 k = 0; k < need; k++)
		if (buf[p0 - 1 - k] != buf[op - 1 - k])
			return false;
	return true;
}

// The dense variant measures every tag hit instead of asking "is it a match at all": the forward extent (as
// single_match_len computes it) and the backward extent BEFORE last_match clips it -- min(equal bytes, p0, op) --, so that
// the in-order pass of a round can price the hit under whatever last_match the emissions in front of it leave
// (len = fwd + min(back, p0 - last_match), < MINIMUM_MATCH = 0; src/rzip.c:431-461).  Packed: fwd in bits 0..15, back in
// 16..30, bit 31 = the backward compare stopped at its cap (the extent may be longer).  ~0 = the forward compare ran
// into its cap: the candidate takes the exact step.
constexpr uint32_t HIT_UNMEASURED = 0xFFFFFFFFu;
__device__ __forceinline__ uint32_t measure_hit(const uint8_t *buf, i64 p0, i64 op, i64 end)
{
	constexpr i64 FWD_CAP = 512, BACK_CAP = 256;
	if (op >= p0)
		return 0;
	i64 total = end - p0;
	if (total < 0)
		total = 0;
	const i64 capf = total < FWD_CAP ? total : FWD_CAP;
	i64 fwd = 0;
	while (fwd + 8 <= capf) {
		const u64 x = reinterpret_cast<const U64u *>(buf + p0 + fwd)->v ^ reinterpret_cast<const U64u *>(buf + op + fwd)->v;
		if (x) {
			fwd += (__ffsll((long long)x) - 1) >> 3;
			goto fwd_done;
		}
		fwd += 8;
	}
	while (fwd < capf && buf[p0 + fwd] == buf[op + fwd])
		fwd++;
	if (fwd == capf && capf < total)
		return HIT_UNMEASURED;
fwd_done:;
	const i64 max_back = op; // (op < p0)
	const i64 capb = max_back < BACK_CAP ? max_back : BACK_CAP;
	i64 back = 0;
	while (back + 8 <= capb) {
		const u64 x = reinterpret_cast<const U64u *>(buf + p0 - 8 - back)->v ^ reinterpret_cast<const U64u *>(buf + op - 8 - back)->v;
		if (x) {
			back += (i64)(__clzll((long long)x) >> 3);
			goto back_done;
		}
		back += 8;
	}
	while (back < capb && buf[p0 - 1 - back] == buf[op - 1 - back])
		back++;
back_done:;
	return (uint32_t)fwd | (uint32_t)back << 16 | ((back == capb && capb < max_back) ? 0x80000000u : 0u);
}

// Equal leading bytes of two 16-byte pieces; equal trailing bytes (the piece's last byte is the one next to the position).
__device__ __forceinline__ int lead16(const U128u &a, const U128u &b)
{
	const u64 x0 = a.a ^ b.a, x1 = a.b ^ b.b;
	return x0 ? (__ffsll((long long)x0) - 1) >> 3 : x1 ? 8 + ((__ffsll((long long)x1) - 1) >> 3) : 16;
}
__device__ __forceinline__ int trail16(const U128u &a, const U128u &b)
{
	const u64 x0 = a.a ^ b.a, x1 = a.b ^ b.b;
	return x1 ? __clzll((long long)x1) >> 3 : x0 ? 8 + (__clzll((long long)x0) >> 3) : 16;
}

// measure_hit for all n hits of a lane, with the loads of four hits in flight together: 32 bytes each way first (a hit
// that ends within them both ways is measured; one that ends within 15 both ways is a certain miss), the next 32 for
// the directions still open, the byte-wise measure_hit beyond 64 (rare) and near the chunk's ends.  One lane's hits were
// measured one after the other until round 6: two dependent trips to memory per hit, sixteen hits per candidate on the
// inputs this variant is for.
__device__ __forceinline__ void measure_hits(const uint8_t *buf, const i64 P, const i64 end, const int n, const int lane, const i64 *hit_lds,
					     uint32_t *hit_fr, bool &pot, bool &unmeasured)
{
	auto note = [&](int k, uint32_t fr) {
		hit_fr[k * 64 + lane] = fr;
		if (fr == HIT_UNMEASURED)
			unmeasured = true;
		else if ((fr & 0xFFFFu) + ((fr >> 16) & 0x7FFFu) >= (uint32_t)MINIMUM_MATCH)
			pot = true;
	};
	auto slow = [&](int k, i64 op) { note(k, quick_reject(buf, P, op, end, 0) ? 0u : measure_hit(buf, P, op, end)); };
	if (P < 64 || end - P < 64) {
		for (int k = 0; k < n; k++)
			slow(k, hit_lds[k * 64 + lane]);
		return;
	}
	U128u pf[4], pb[4]; // the candidate's own 64 bytes each way
#pragma unroll
	for (int c = 0; c < 4; c++) {
		pf[c] = *reinterpret_cast<const U128u *>(buf + P + 16 * c);
		pb[c] = *reinterpret_cast<const U128u *>(buf + P - 16 - 16 * c);
	}
	u64 open = 0; // hits with a direction that ran through its first 32 bytes
	for (int k0 = 0; k0 < n; k0 += 4) {
		i64 op[4];
		U128u f[4][2], b[4][2];
		bool fast[4];
#pragma unroll
		for (int j = 0; j < 4; j++) {
			op[j] = k0 + j < n ? hit_lds[(k0 + j) * 64 + lane] : P;
			fast[j] = op[j] >= 64 && op[j] < P;
			if (fast[j]) {
#pragma unroll
				for (int c = 0; c < 2; c++) {
					f[j][c] = *reinterpret_cast<const U128u *>(buf + op[j] + 16 * c);
					b[j][c] = *reinterpret_cast<const U128u *>(buf + op[j] - 16 - 16 * c);
				}
			}
		}
#pragma unroll
		for (int j = 0; j < 4; j++) {
			if (k0 + j >= n)
				continue;
			if (op[j] >= P)
				note(k0 + j, 0);
			else if (!fast[j])
				slow(k0 + j, op[j]);
			else {
				int fw = lead16(pf[0], f[j][0]), bk = trail16(pb[0], b[j][0]);
				if (fw == 16)
					fw += lead16(pf[1], f[j][1]);
				if (bk == 16)
					bk += trail16(pb[1], b[j][1]);
				if (fw == 32 || bk == 32) {
					open |= 1ull << (k0 + j);
					hit_fr[(k0 + j) * 64 + lane] = (uint32_t)fw | (uint32_t)bk << 16; // (so far)
				} else
					note(k0 + j, (uint32_t)fw | (uint32_t)bk << 16);
			}
		}
	}
	while (open) {
		int kk[4];
		i64 op[4];
		int fw[4], bk[4];
		U128u f[4][2], b[4][2];
#pragma unroll
		for (int j = 0; j < 4; j++) {
			kk[j] = open ? __ffsll((long long)open) - 1 : -1;
			if (kk[j] >= 0) {
				open &= open - 1;
				op[j] = hit_lds[kk[j] * 64 + lane];
				const uint32_t fr = hit_fr[kk[j] * 64 + lane];
				fw[j] = (int)(fr & 0xFFFFu);
				bk[j] = (int)(fr >> 16);
#pragma unroll
				for (int c = 0; c < 2; c++) {
					if (fw[j] == 32)
						f[j][c] = *reinterpret_cast<const U128u *>(buf + op[j] + 32 + 16 * c);
					if (bk[j] == 32)
						b[j][c] = *reinterpret_cast<const U128u *>(buf + op[j] - 48 - 16 * c);
				}
			}
		}
#pragma unroll
		for (int j = 0; j < 4; j++) {
			if (kk[j] < 0)
				continue;
			if (fw[j] == 32) {
				fw[j] += lead16(pf[2], f[j][0]);
				if (fw[j] == 48)
					fw[j] += lead16(pf[3], f[j][1]);
			}
			if (bk[j] == 32) {
				bk[j] += trail16(pb[2], b[j][0]);
				if (bk[j] == 48)
					bk[j] += trail16(pb[3], b[j][1]);
			}
			if (fw[j] == 64 || bk[j] == 64)
				note(kk[j], measure_hit(buf, P, op[j], end)); // longer than 64 one way: byte-wise from the start
			else
				note(kk[j], (uint32_t)fw[j] | (uint32_t)bk[j] << 16);
		}
	}
}

// Phase A/B of a resolver round, for the lanes with need_sim: one automaton step each, simulated
// against the table as it stands.  Wave-convergent, so that every dependent HBM round trip is shared
// by all simulating lanes: (A1) lookup walk in 64-slot steps over the side arrays, (A2) verification of
// tag hits, (A3) the displacement chain of the insert, one level at a time.  Called by the resolver
// wave for its window and by the pre-simulation waves for the candidates still in the queue; R is
// only read (masks, table pointers, last_match); hit_lds / eqs_lds are the caller's own LDS scratch.
template <int MAXH, int MAXE, bool DENSE, class LapF>
__device__ __forceinline__ void simulate_lanes(const Resolver &R, const Slot *__restrict__ tbl, const uint8_t *__restrict__ buf, const i64 tbl_size,
					       const u64 better, const int lane, const bool alive, const bool need_sim, const u64 w_tag,
					       const i64 w_pos, const int w_ticket, i64 *hit_lds, uint32_t *eqs_lds, const int eqs_stride, LaneSim &L, LapF lap,
					       uint32_t *hit_fr = nullptr, u64 *q_cache = nullptr)
{
		const u64 T = w_tag;
		const i64 P = w_pos;
		const int my_rank = bitness_rank(T);
		const int nb1 = __popcll(better) + 1; // rank bytes below this are due for cleaning
		int kind = -1; // insert stop: 0 empty, 1 due for cleaning, 2 lesser bitness
		i64 sidx = 0;
		Slot occ;
		occ.offset = 0;
		occ.t = 0;
		int nhit = 0; // tag hits met by the lookup walk, offsets parked in LDS (hit_lds)
		// Twins: consecutive candidates with the SAME tag (the byte leaving the 31-byte window
		// equals the byte entering it) share a bucket, so the second always conflicts with the
		// first one's insert.  The second of such a pair is simulated on top of the insert it can
		// predict for the first from its own walk (same tag, same table); phase C checks the prediction.
		// (cross-lane reads are made by all lanes: no short-circuit evaluation around them)
		const int prev_alive = __shfl_up((int)alive, 1);
		const u64 prev_tag = bcast64(w_tag, (lane + 63) & 63);
		const i64 P_prev = (i64)bcast64((u64)w_pos, (lane + 63) & 63);
		const bool tw_cand = lane > 0 && alive && prev_alive != 0 && prev_tag == w_tag;
		const int prev_cand = __shfl_up((int)tw_cand, 1);
		bool tw = tw_cand && prev_cand == 0; // a third twin in a row takes the conflict path
		bool seek_pred = false, tw_hit = false;
		// ---- A1: lookup walk to the first empty slot, 16 slots per step, branch-free masks ----
		{
			i64 idx = (i64)(T & R.hmask);
			uint32_t neq = 0;
			int steps = 0;
			bool fin = !need_sim;
			const int thr = my_rank > nb1 ? my_rank : nb1; // ranks below this stop my insert
			if (need_sim) {
				L.complex_ = nb1 >= 63 || my_rank >= 63; // rank bytes saturate at 63
				L.match = false;
				L.victim = false;
				L.dec = 0;
				L.misses = 0;
				L.nw = 0;
				L.ins = (T & R.tag_mask) == R.tag_mask;
				L.lo = (uint32_t)idx;
				L.hi = (uint32_t)idx;
				if (L.complex_)
					fin = true;
				L.tw_over = false;
				L.soft1 = false;
				tw = tw && L.ins && !L.complex_;
				seek_pred = tw;
			}
			const u64 thr8 = (u64)thr * B01, fp8 = (u64)fp_byte(T) * B01;
			while (__ballot(!fin)) {
				if (!fin) {
					// 64 slots per step: their rank and fingerprint bytes (the arrays are padded)
					U128u r4[4], f4[4];
#pragma unroll
					for (int c = 0; c < 4; c++) {
						r4[c] = *reinterpret_cast<const U128u *>(R.rk + idx + 16 * c);
						f4[c] = *reinterpret_cast<const U128u *>(R.fpa + idx + 16 * c);
					}
					u64 E = 0, S = 0, Q = 0; // bit q: slot idx + q is empty / stops my insert / may hold my tag
#pragma unroll
					for (int c = 0; c < 4; c++) {
						E |= (u64)((flags_to_bits(bytes_lt(r4[c].b, B01)) << 8) | flags_to_bits(bytes_lt(r4[c].a, B01))) << (16 * c);
						S |= (u64)((flags_to_bits(bytes_lt(r4[c].b, thr8)) << 8) | flags_to_bits(bytes_lt(r4[c].a, thr8))) << (16 * c);
						Q |= (u64)((flags_to_bits(bytes_eq(f4[c].b, fp8)) << 8) | flags_to_bits(bytes_eq(f4[c].a, fp8))) << (16 * c);
						// loose masks mean short clusters: stop evaluating once every walking lane has met
						// its first empty slot (two for a twin, whose lookup may walk on past the first)
						if (c < 3 && !__ballot(tw ? __popcll(E) < 2 : E == 0))
							break;
					}
					steps += 64;
					if (idx + 64 > tbl_size || steps > A1_MAX_STEPS) {
						L.complex_ = true;
						L.hi = (uint32_t)(idx + 63 < tbl_size - 1 ? idx + 63 : tbl_size - 1);
						fin = true;
					} else {
						// The slots whose fingerprint matches are looked at twice below (equal tags in front of the insert's
						// stop, tag hits in front of the first empty slot), one dependent load after the other: the first three
						// of them are fetched here, together -- one round trip instead of up to six (a fourth and later
						// one, rare, is loaded where it is needed)
						int q0 = -1, q1 = -1, q2 = -1;
						u64 c0t = 0, c1t = 0, c2t = 0;
						i64 c0o = 0, c1o = 0, c2o = 0;
						{
							u64 Qp = Q;
							if (Qp) {
								q0 = __ffsll((long long)Qp) - 1;
								Qp &= Qp - 1;
								const Slot c = tbl[idx + q0];
								c0t = c.t;
								c0o = c.offset;
							}
							if (Qp) {
								q1 = __ffsll((long long)Qp) - 1;
								Qp &= Qp - 1;
								const Slot c = tbl[idx + q1];
								c1t = c.t;
								c1o = c.offset;
							}
							if (Qp) {
								q2 = __ffsll((long long)Qp) - 1;
								Qp &= Qp - 1;
								const Slot c = tbl[idx + q2];
								c2t = c.t;
								c2o = c.offset;
							}
							if constexpr (DENSE) {
								// full chains (sixteen equal tags in a row at level 7): the next sixteen fingerprint matches
								// of the step are fetched together as well -- unconditional loads, so that they are all on
								// their way before the first is waited for -- and parked in LDS (q_cache[2 j], [2 j + 1])
								if (__ballot(Qp != 0)) {
									Slot c[16];
									int ql = q2 < 0 ? 0 : q2;
#pragma unroll
									for (int j = 0; j < 16; j++) {
										if (Qp) {
											ql = __ffsll((long long)Qp) - 1;
											Qp &= Qp - 1;
										}
										c[j] = tbl[idx + ql];
									}
#pragma unroll
									for (int j = 0; j < 16; j++) {
										q_cache[(2 * j) * 64 + lane] = c[j].t;
										q_cache[(2 * j + 1) * 64 + lane] = (u64)c[j].offset;
									}
								}
							}
						}
						auto slot_at = [&](int q, int word) -> u64 {
							if constexpr (DENSE) {
								const int ord = __popcll(Q & low_mask(q)) - 3; // (the first three are in registers)
								if (ord < 16)
									return q_cache[(2 * ord + word) * 64 + lane];
							}
							return word ? (u64)tbl[idx + q].offset : tbl[idx + q].t;
						};
						auto tag_at = [&](int q) -> u64 { return q == q0 ? c0t : q == q1 ? c1t : q == q2 ? c2t : slot_at(q, 0); };
						auto off_at = [&](int q) -> i64 { return q == q0 ? c0o : q == q1 ? c1o : q == q2 ? c2o : (i64)slot_at(q, 1); };
						u64 Em = E;
						if (kind < 0 && L.ins) {
							int from = 0;
							for (int pass = 0; pass < 2; pass++) {
								const u64 Sm = S & ~low_mask(from);
								const int s1 = Sm ? __ffsll((long long)Sm) - 1 : 64;
								u64 eqb = Q & low_mask(s1) & ~low_mask(from);
								while (eqb) {
									const int q = __ffsll((long long)eqb) - 1;
									eqb &= eqb - 1;
									if (tag_at(q) != T)
										continue; // fingerprint false positive
									if (neq < MAXE)
										eqs_lds[neq * eqs_stride + w_ticket] = (uint32_t)(idx + q);
									if (++neq >= R.max_chain) {
										if (R.max_chain <= MAXE && !tw)
											kind = 3; // round-robin eviction among these equal tags
										else if (seek_pred) {
											tw = seek_pred = false;
											if (R.max_chain <= MAXE)
												kind = 3;
											else
												L.complex_ = true;
										} else
											L.complex_ = true;
										eqb = 0;
									}
								}
								if (kind >= 0 || L.complex_ || s1 >= 64)
									break;
								// the stop's rank byte, from the registers of this step (selects: an indexed array would go to scratch)
								const u64 rw = s1 < 8 ? r4[0].a : s1 < 16 ? r4[0].b : s1 < 24 ? r4[1].a : s1 < 32 ? r4[1].b : s1 < 40 ? r4[2].a : s1 < 48 ? r4[2].b : s1 < 56 ? r4[3].a : r4[3].b;
								const int rv = (int)((rw >> (8 * (s1 & 7))) & 0xFF); // which kind of stop is it?
								const int k1 = rv == 0 ? 0 : rv < nb1 ? 1 : 2;
								if (!seek_pred) {
									kind = k1;
									sidx = idx + s1; // a lesser-bitness occupant (kind 2) is fetched after the walk
									if (DENSE && k1 == 1 && ((Q >> s1) & 1) && tag_at(s1) == T)
										L.soft1 = true;
									break; // (a twin that displaces in its turn: A3 checks its walks against tw_slot)
								}
								seek_pred = false;
								if (k1 == 1 && ((Q >> s1) & 1)) {
									tw = false; // the predecessor may replace its own tag: conflict path
									kind = k1;
									sidx = idx + s1;
									if (DENSE && tag_at(s1) == T)
										L.soft1 = true;
									break;
								}
								L.tw_slot = (uint32_t)(idx + s1);
								L.tw_kind = k1; // 2: the predecessor takes the slot of a lesser-bitness occupant, which it
										// re-inserts elsewhere -- a foreign write like any other for phase C
								tw_hit = true;
								if (k1 == 0)
									Em &= ~(1ull << s1); // the twin fills the first empty slot: my lookup walks on
								if ((T & better) != better) {
									// before the first clean of a chunk a tag may itself be due for cleaning: the
									// predecessor's entry is then the stop of my own insert, which replaces it
									kind = 1;
									sidx = idx + s1;
									L.tw_over = true;
									break;
								}
								if (neq < MAXE)
									eqs_lds[neq * eqs_stride + w_ticket] = (uint32_t)(idx + s1);
								if (++neq >= R.max_chain) {
									L.complex_ = true;
									break;
								}
								from = s1 + 1;
							}
						}
						const int fe = Em ? __ffsll((long long)Em) - 1 : 64; // first empty slot of the step
						u64 hm = Q & low_mask(fe);
						while (hm) {
							const int q = __ffsll((long long)hm) - 1;
							hm &= hm - 1;
							if (tag_at(q) != T)
								continue; // fingerprint false positive
							if (nhit < MAXH)
								hit_lds[nhit * 64 + lane] = off_at(q);
							else
								L.complex_ = true;
							nhit++;
						}
						if (fe < 64) {
							L.hi = (uint32_t)(idx + fe);
							fin = true;
						}
					}
					idx += 64;
				}
			}
		}

		if (need_sim) {
			L.k0 = kind;
			if (kind == 2 && !L.complex_)
				occ = tbl[sidx]; // just read by the walk: an L2 hit, and only displacing inserts pay for it
		}
		lap(9);
		// ---- A2: are the tag hits real matches (>= MINIMUM_MATCH)? ----
		if constexpr (DENSE) {
			// every hit measured (measure_hit): what it is worth is decided in candidate order, under the last_match
			// the emissions in front of the candidate leave behind
			if (need_sim) {
				L.nh = 0;
				L.pot = false;
			}
			if (need_sim && !L.complex_) {
				const int n = nhit < MAXH ? nhit : MAXH;
				bool pot = false, unmeasured = false;
				measure_hits(buf, P, R.end, n, lane, hit_lds, hit_fr, pot, unmeasured);
				if (unmeasured)
					L.complex_ = true;
				L.nh = n;
				L.pot = pot;
				L.match = pot;
				L.misses = n; // (of a candidate without a possible match: every hit a miss under any last_match)
			}
		} else {
			// all hits are pre-tested with independent loads (one round trip), the rare
			// undecided ones get the exact compare
			uint32_t und = 0; // bit k: hit k needs the exact compare
			if (need_sim && !L.complex_) {
				for (int k = 0; k < nhit && k < MAXH; k++)
					if (!quick_reject(buf, P, hit_lds[k * 64 + lane], R.end, R.last_match))
						und |= 1u << k;
				L.misses = (nhit < MAXH ? nhit : MAXH) - __popc(und);
			}
			while (__ballot(und != 0)) {
				if (und) {
					const int k = __ffs((int)und) - 1;
					und &= und - 1;
					if (!L.match) {
						if (lane_verify(buf, P, hit_lds[k * 64 + lane], R.end, R.last_match))
							L.match = true;
						else
							L.misses++;
					}
				}
			}
		}
		// the twin's entry is one more tag hit for its successor: a real match (a run of one byte
		// value) goes to the serial path
		if (__ballot(need_sim && tw && tw_hit && !L.complex_)) {
			if (need_sim && tw && tw_hit && !L.complex_) {
				// (independent 8-byte looks first: text twins differ right there)
				if (!quick_reject(buf, P, P_prev, R.end, R.last_match) && lane_verify(buf, P, P_prev, R.end, R.last_match))
					L.complex_ = true;
				else
					L.misses++;
			}
		}
		if (need_sim)
			L.twin = tw && tw_hit && !L.complex_;
		lap(14);
		// ---- A3: displacement chain of the insert, level by level ----
		{
			u64 cur_t = T;
			i64 cur_off = P;
			bool chain = need_sim && L.ins && !L.complex_ && (DENSE || !L.match); // (dense: a match does not end the round)
#pragma unroll 1
			for (int level = 0; level < 5; level++) {
				if (!__ballot(chain))
					break;
				bool walk = false;
				i64 j = 0;
				int r2 = 0;
				i64 home2 = 0;
				if (chain) {
					if (L.nw == 4) {
						L.complex_ = true;
						chain = false;
					} else {
#pragma unroll
						for (int k = 0; k < 4; k++)
							if (k == L.nw) {
								L.w_slot[k] = (uint32_t)sidx;
								L.w_t[k] = cur_t;
								L.w_off[k] = cur_off;
							}
						L.nw++;
						if ((uint32_t)sidx > L.hi)
							L.hi = (uint32_t)sidx;
						if (kind == 0) {
							chain = false;
						} else if (kind == 1) {
							L.dec = 1;
							chain = false;
						} else if (kind == 3) {
							// victim slot depends on victim_round at commit time (set in phase C)
							L.victim = true;
							L.dec = 1;
							chain = false;
						} else {
							// lesser-bitness occupant: it is re-inserted from its own bucket
							cur_t = occ.t;
							cur_off = occ.offset;
							r2 = bitness_rank(cur_t);
							j = (i64)(cur_t & R.hmask);
							home2 = j;
							if ((uint32_t)j < L.lo)
								L.lo = (uint32_t)j;
							kind = -1;
							walk = true;
						}
					}
				}
				uint32_t neq2 = 0;
				int st2 = 0;
				const int thr2 = r2 > nb1 ? r2 : nb1;
				const u64 thr2_8 = (u64)thr2 * B01, fp2_8 = (u64)fp_byte(cur_t) * B01;
				while (__ballot(walk)) {
					if (walk) {
						U128u r4[4], f4[4];
#pragma unroll
						for (int c = 0; c < 4; c++) {
							r4[c] = *reinterpret_cast<const U128u *>(R.rk + j + 16 * c);
							f4[c] = *reinterpret_cast<const U128u *>(R.fpa + j + 16 * c);
						}
						u64 S2 = 0, Q2 = 0;
#pragma unroll
						for (int c = 0; c < 4; c++) {
							S2 |= (u64)((flags_to_bits(bytes_lt(r4[c].b, thr2_8)) << 8) | flags_to_bits(bytes_lt(r4[c].a, thr2_8))) << (16 * c);
							Q2 |= (u64)((flags_to_bits(bytes_eq(f4[c].b, fp2_8)) << 8) | flags_to_bits(bytes_eq(f4[c].a, fp2_8))) << (16 * c);
							if (c < 3 && !__ballot(S2 == 0))
								break; // every walking lane has its stop
						}
						st2 += 64;
						if (j + 64 > tbl_size || st2 > A1_MAX_STEPS || r2 >= 63) {
							L.complex_ = true;
							chain = false;
							walk = false;
						} else {
							const int s2 = S2 ? __ffsll((long long)S2) - 1 : 64;
							u64 eq2 = Q2 & low_mask(s2);
							while (eq2) {
								const int q = __ffsll((long long)eq2) - 1;
								eq2 &= eq2 - 1;
								if (tbl[j + q].t == cur_t && ++neq2 >= R.max_chain) {
									L.complex_ = true;
									chain = false;
									walk = false;
									eq2 = 0;
								}
							}
							if (walk && s2 < 64) {
								sidx = j + s2;
								const int rv = R.rk[sidx];
								kind = rv == 0 ? 0 : rv < nb1 ? 1 : 2;
								if (kind == 2)
									occ = tbl[sidx];
								walk = false;
								// a twin's re-insert walk ran over the table WITHOUT its predecessor's insert:
								// exact only if it never met that slot
								if (tw && tw_hit && (i64)L.tw_slot >= home2 && (i64)L.tw_slot <= sidx) {
									L.complex_ = true;
									chain = false;
								}
							}
						}
						j += 64;
					}
				}
			}
			if (chain)
				L.complex_ = true; // deeper than the write list allows
		}
}
#include "rzip_resolve_mw.h"

// ---------------------------------------------------------------------------------------------
// K3: grid-wide forward extent of one long match
// ---------------------------------------------------------------------------------------------
// Equal bytes of buf[p + start ..] and buf[op + start ..] (op < p), at most `limit` in total, for
// a match the resolver wave found still equal after LONG_EXTENT bytes (a single wave compares at
// ~1 GB/s; the whole chip streams the two operands at HBM speed).  *best holds the smallest
// mismatch offset found so far (initialised to `limit`); tiles past it are skipped.
constexpr int LC_TILE = 1 << 16;
__global__ void __launch_bounds__(256) k_long_compare(const uint8_t *__restrict__ buf, i64 p, i64 op, i64 start, i64 limit,
						      unsigned long long *__restrict__ best)
{
	const i64 ntiles = (limit - start + LC_TILE - 1) / LC_TILE;
	for (i64 t = blockIdx.x; t < ntiles; t += gridDim.x) {
		const i64 t0 = start + t * LC_TILE;
		if ((unsigned long long)t0 >= __hip_atomic_load(best, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
			return; // tiles only grow from here
		const i64 t1 = t0 + LC_TILE < limit ? t0 + LC_TILE : limit;
		i64 mine = (i64)1 << 62;
		for (i64 off = t0 + (i64)threadIdx.x * 16; off < t1; off += 256 * 16) {
			if (off + 16 <= t1) {
				const U128u a = *reinterpret_cast<const U128u *>(buf + p + off);
				const U128u b = *reinterpret_cast<const U128u *>(buf + op + off);
				const u64 x0 = a.a ^ b.a, x1 = a.b ^ b.b;
				if (x0 | x1) {
					mine = off + (x0 ? (__ffsll((long long)x0) - 1) >> 3 : 8 + ((__ffsll((long long)x1) - 1) >> 3));
					break;
				}
			} else {
				for (i64 k = off; k < t1; k++)
					if (buf[p + k] != buf[op + k]) {
						mine = k;
						break;
					}
				break;
			}
		}
		if (mine < ((i64)1 << 62))
			atomicMin(best, (unsigned long long)mine);
	}
}

// ---------------------------------------------------------------------------------------------
// K4: literal gather (stream 1)
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_gather_runs(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst,
						     const CopyRun *__restrict__ runs, int nruns, i64 dst_lo, i64 dst_hi)
{
	// each workgroup owns one 64 KiB-aligned tile of the destination, clipped to [dst_lo, dst_hi)
	const i64 tile0 = (dst_lo & ~(i64)65535) + (i64)blockIdx.x * 65536;
	i64 tile1 = tile0 + 65536;
	if (tile1 > dst_hi)
		tile1 = dst_hi;
	// binary search: last run with dst_off <= tile0
	int lo = 0, hi = nruns - 1;
	while (lo < hi) {
		int mid = (lo + hi + 1) >> 1;
		if (runs[mid].dst_off <= tile0)
			lo = mid;
		else
			hi = mid - 1;
	}
	int r = lo;
	for (i64 d = tile0 + (i64)threadIdx.x * 16; d < tile1; d += 256 * 16) {
		while (r + 1 < nruns && runs[r + 1].dst_off <= d)
			r++;
		CopyRun cr = runs[r];
		i64 in_run = d - cr.dst_off;
		if (d >= dst_lo && in_run >= 0 && in_run + 16 <= cr.len && d + 16 <= tile1) {
			U128u v = *reinterpret_cast<const U128u *>(src + cr.src_off + in_run);
			uint4 o;
			o.x = (uint32_t)v.a;
			o.y = (uint32_t)(v.a >> 32);
			o.z = (uint32_t)v.b;
			o.w = (uint32_t)(v.b >> 32);
			*reinterpret_cast<uint4 *>(dst + d) = o;
		} else {
			int rr = r;
			for (int k = 0; k < 16 && d + k < tile1; k++) {
				i64 dd = d + k;
				if (dd < dst_lo)
					continue;
				while (rr + 1 < nruns && runs[rr + 1].dst_off <= dd)
					rr++;
				const i64 ir = dd - runs[rr].dst_off;
				if (ir >= 0 && ir < runs[rr].len)
					dst[dd] = src[runs[rr].src_off + ir];
			}
		}
	}
}

int gather_runs_device(const uint8_t *d_src, uint8_t *d_dst, const CopyRun *d_runs, int nruns, int64_t dst_lo, int64_t dst_hi, hipStream_t s)
{
	if (dst_hi <= dst_lo || nruns <= 0)
		return 0;
	const int64_t first = dst_lo & ~(int64_t)65535;
	int64_t tiles = (dst_hi - first + 65535) / 65536;
	hipLaunchKernelGGL(k_gather_runs, dim3((unsigned)tiles), dim3(256), 0, s, d_src, d_dst, d_runs, nruns, (i64)dst_lo, (i64)dst_hi);
	return hipGetLastError() == hipSuccess ? 0 : -1;
}

// ---------------------------------------------------------------------------------------------
// K5: CRC-32 (IEEE, reflected).  raw CRC (init 0, no final xor) per 64 KiB tile; tiles are folded
// on the host with the x^(8*65536) operator.
// ---------------------------------------------------------------------------------------------
constexpr int CRC_TILE = 65536;

__device__ __forceinline__ uint32_t gf2_apply(const uint32_t *mat, uint32_t v)
{
	uint32_t r = 0;
#pragma unroll
	for (int b = 0; b < 32; b++)
		r ^= mat[b] & (0u - ((v >> b) & 1));
	return r;
}

// ops: 8 matrices (32 u32 each): shift by 256 << k bytes, k = 0..7
__global__ void __launch_bounds__(256) k_crc32_tiles(const uint8_t *__restrict__ buf, i64 n, const uint32_t *__restrict__ ops,
						     uint32_t *__restrict__ partial)
{
	__shared__ uint32_t tab[256];
	__shared__ uint32_t mats[8 * 32];
	__shared__ __attribute__((aligned(16))) uint8_t tile[CRC_TILE];
	__shared__ uint32_t red[256];
	const int tid = threadIdx.x;
	{
		uint32_t r = tid;
		for (int j = 0; j < 8; j++)
			r = (r >> 1) ^ (0xEDB88320u & (0u - (r & 1)));
		tab[tid] = r;
		mats[tid] = ops[tid];
	}
	const i64 t0 = (i64)blockIdx.x * CRC_TILE;
	i64 len = n - t0;
	if (len > CRC_TILE)
		len = CRC_TILE;
	// stage (zero-padded FRONT so that every thread covers exactly 256 bytes: leading zeros do not
	// change a raw CRC with init 0)
	const i64 pad = CRC_TILE - len;
	for (int off = tid * 16; off < CRC_TILE; off += 256 * 16) {
		uint4 v = make_uint4(0, 0, 0, 0);
		i64 s = (i64)off - pad; // source offset within the tile's data
		if (s >= 0 && s + 16 <= len && (((uintptr_t)(buf + t0 + s)) & 15) == 0)
			v = *reinterpret_cast<const uint4 *>(buf + t0 + s);
		else {
			uint8_t *vb = reinterpret_cast<uint8_t *>(&v);
			for (int k = 0; k < 16; k++) {
				i64 ss = s + k;
				vb[k] = (ss >= 0 && ss < len) ? buf[t0 + ss] : 0;
			}
		}
		*reinterpret_cast<uint4 *>(tile + off) = v;
	}
	__syncthreads();
	uint32_t c = 0;
	const uint8_t *mine = tile + tid * 256;
	for (int k = 0; k < 256; k += 4) {
		uint32_t w = *reinterpret_cast<const uint32_t *>(mine + k);
		c ^= w;
		c = tab[c & 0xFF] ^ (c >> 8);
		c = tab[c & 0xFF] ^ (c >> 8);
		c = tab[c & 0xFF] ^ (c >> 8);
		c = tab[c & 0xFF] ^ (c >> 8);
	}
	red[tid] = c;
	__syncthreads();
	// tree fold: combine(left, right) = shift(left, len(right)) ^ right
	for (int k = 0; k < 8; k++) {
		int stride = 1 << k;
		if ((tid & (2 * stride - 1)) == 0)
			red[tid] = gf2_apply(mats + k * 32, red[tid]) ^ red[tid + stride];
		__syncthreads();
	}
	if (tid == 0)
		partial[blockIdx.x] = red[0];
}

static void gf2_square(uint32_t *sq, const uint32_t *m)
{
	for (int i = 0; i < 32; i++) {
		uint32_t v = m[i], r = 0;
		for (int b = 0; v; b++, v >>= 1)
			if (v & 1)
				r ^= m[b];
		sq[i] = r;
	}
}
static uint32_t gf2_times(const uint32_t *m, uint32_t v)
{
	uint32_t r = 0;
	for (int b = 0; v; b++, v >>= 1)
		if (v & 1)
			r ^= m[b];
	return r;
}
// operator "append one zero byte" for the reflected CRC-32 register, then powers of two of it
static void crc_shift_ops(uint32_t ops[64][32])
{
	uint32_t bit[32], t[32];
	// one zero BIT: v -> (v >> 1) ^ (poly if v & 1)
	bit[0] = 0xEDB88320u;
	for (int i = 1; i < 32; i++)
		bit[i] = 1u << (i - 1);
	gf2_square(t, bit);   // 2 bits
	gf2_square(bit, t);   // 4 bits
	gf2_square(ops[0], bit); // 8 bits = 1 byte
	for (int k = 1; k < 64; k++)
		gf2_square(ops[k], ops[k - 1]); // 2^k bytes
}
static uint32_t crc_shift(const uint32_t ops[64][32], uint32_t v, uint64_t nbytes)
{
	for (int k = 0; nbytes; k++, nbytes >>= 1)
		if (nbytes & 1)
			v = gf2_times(ops[k], v);
	return v;
}

int crc32_device(ScanWorkspace *w, const uint8_t *d_buf, int64_t n, uint32_t *crc, hipStream_t s)
{
	static uint32_t ops[64][32];
	static bool ready = false;
	if (!ready) {
		crc_shift_ops(ops);
		ready = true;
	}
	if (n <= 0) {
		*crc = 0;
		return 0;
	}
	size_t tiles = (size_t)((n + CRC_TILE - 1) / CRC_TILE);
	if (tiles > w->crc_cap)
		return -2;
	uint32_t *d_ops = w->crc_partial + w->crc_cap; // 256 u32 reserved after the partials
	HIPCHK(hipMemcpyAsync(d_ops, &ops[8][0], 8 * 32 * 4, hipMemcpyHostToDevice, s)); // 256 B << k
	EventTimer tc(s);
	hipLaunchKernelGGL(k_crc32_tiles, dim3((unsigned)tiles), dim3(256), 0, s, d_buf, (i64)n, d_ops, w->crc_partial);
	tc.stop();
	std::vector<uint32_t> part(tiles);
	HIPCHK(d2h_pageable(part.data(), w->crc_partial, tiles * 4, s));
	{
		ProfileStore &ps = ProfileStore::get();
		std::lock_guard<std::mutex> lk(ps.mu);
		ps.p.crc_ms += tc.ms_noted(ps, PK_CRC);
		ps.p.crc_launches++;
		ps.p.crc_bytes += n;
	}
	// tiles 0..T-2 are full; the last one was front-padded with zeros, i.e. its raw CRC is that of
	// its real bytes.  raw(total) = fold over tiles with the right shift lengths.
	uint32_t raw = 0;
	for (size_t t = 0; t < tiles; t++) {
		int64_t len = (t + 1 < tiles) ? CRC_TILE : n - (int64_t)t * CRC_TILE;
		raw = crc_shift(ops, raw, (uint64_t)len) ^ part[t];
	}
	*crc = raw ^ crc_shift(ops, 0xFFFFFFFFu, (uint64_t)n) ^ 0xFFFFFFFFu;
	return 0;
}

// K1 on its own (lrzgpu_tag_candidates_dev: parity of the tags against the scan access hooks, and the kernel's rate
// alone on the GPU): count and an order-independent checksum of the (position, tag) pairs of the per-tile lists.
__global__ void __launch_bounds__(256) k_cand_checksum(const uint32_t *__restrict__ cand_rel, const u64 *__restrict__ cand_tag,
						       const uint32_t *__restrict__ tile_count, i64 seg_lo, i64 first, unsigned long long *__restrict__ out)
{
	const uint32_t cnt = tile_count[blockIdx.x];
	const size_t in0 = (size_t)blockIdx.x * TILE;
	u64 sum = 0, n = 0;
	for (uint32_t i = threadIdx.x; i < cnt; i += 256) {
		const i64 pos = seg_lo + (i64)cand_rel[in0 + i];
		if (pos >= first) {
			sum += ((u64)pos + 1) * 0x9E3779B97F4A7C15ull ^ cand_tag[in0 + i] * 0xC2B2AE3D27D4EB4Full;
			n++;
		}
	}
#pragma unroll
	for (int d = 32; d >= 1; d >>= 1) {
		sum += bcast64(sum, (threadIdx.x & 63) ^ d);
		n += bcast64(n, (threadIdx.x & 63) ^ d);
	}
	if ((threadIdx.x & 63) == 0 && n) {
		atomicAdd(&out[0], (unsigned long long)n);
		atomicAdd(&out[1], (unsigned long long)sum);
	}
}

// The candidates of positions [first, end] of a chunk under `min_mask`, segment by segment like the scan; the K1 trio is
// run `reps` times (events around all of it).  d_out: two 64-bit words, zeroed here.
int tag_candidates_device(ScanWorkspace *w, const uint8_t *d_chunk, int64_t first, int64_t end, uint64_t min_mask, int reps,
			  unsigned long long *d_out, double *ms, hipStream_t s, bool only_tags)
{
	ScanState h;
	memset(&h, 0, sizeof(h));
	h.min_mask = h.tag_mask = min_mask;
	HIPCHK(hipMemcpyAsync(w->state, &h, sizeof(h), hipMemcpyHostToDevice, s));
	double total_ms = 0;
	HIPCHK(hipMemsetAsync(d_out, 0, 16, s));
	for (int r = 0; r < (reps < 1 ? 1 : reps); r++) {
		EventTimer t(s);
		for (int64_t lo = first & ~(int64_t)15; lo <= end;) {
			int64_t hi = lo + (int64_t)w->seg_cap - TILE;
			if (hi > end + 1)
				hi = end + 1;
			const int ntiles = (int)((hi - lo + TILE - 1) / TILE);
			hipLaunchKernelGGL(k_tag_scan, dim3(ntiles), dim3(256), 0, s, d_chunk, (i64)lo, (i64)hi, (const u64 *)w->hx, (const ScanState *)w->state,
					   w->cand_rel, (u64 *)w->cand_tag, w->tile_count);
			if (r == 0 || !only_tags) {
				hipLaunchKernelGGL(k_tile_scan, dim3(1), dim3(1024), 0, s, (const uint32_t *)w->tile_count, ntiles, w->tile_base);
				hipLaunchKernelGGL(k_compact_cands, dim3(ntiles), dim3(256), 0, s, (const uint32_t *)w->cand_rel, (const u64 *)w->cand_tag,
						   (const uint32_t *)w->tile_count, (const uint32_t *)w->tile_base, (uint32_t)w->comp_cap, w->comp_rel, (u64 *)w->comp_tag);
			}
			if (r == 0)
				hipLaunchKernelGGL(k_cand_checksum, dim3(ntiles), dim3(256), 0, s, (const uint32_t *)w->cand_rel, (const u64 *)w->cand_tag,
						   (const uint32_t *)w->tile_count, (i64)lo, (i64)first, d_out);
			lo = hi;
		}
		t.stop();
		HIPCHK(stream_wait(s));
		if (r > 0 || reps <= 1)
			total_ms += t.ms();
	}
	if (ms)
		*ms = total_ms / (reps > 1 ? reps - 1 : 1); // (the first pass also runs the checksum: not timed when there are more)
	return 0;
}

// ---------------------------------------------------------------------------------------------
// host orchestration
// ---------------------------------------------------------------------------------------------
int scan_workspace_create(ScanWorkspace **out, int rzip_level, int64_t max_chunk)
{
	ScanWorkspace *w = (ScanWorkspace *)calloc(1, sizeof(ScanWorkspace));
	if (!w)
		return -1;
	unsigned mb, freq, chain;
	rzip_level_params(rzip_level, &mb, &freq, &chain);
	int64_t hashsize = (int64_t)mb * (1048576 / 16);
	for (w->hash_bits = 0; ((int64_t)1 << w->hash_bits) < hashsize; w->hash_bits++)
		;
	HIPCHK(hipMalloc(&w->table, ((size_t)16 << w->hash_bits) + 64 * 16));
	HIPCHK(hipMalloc(&w->rank_bytes, ((size_t)1 << w->hash_bits) + 256));
	HIPCHK(hipMalloc(&w->fp_bytes, ((size_t)1 << w->hash_bits) + 256));
	HIPCHK(hipMalloc(&w->state, sizeof(ScanState)));
	HIPCHK(hipMalloc(&w->hx, 256 * 8));
	{
		w->batch_mode = 1; // (bit 0: speculative batches; 0 = exact serial steps only, the debugging mode of round 1)
		const char *pr = getenv("LRZGPU_TRACE"); // bit 1: per-phase cycle counters in the profile (trace level 3)
		if (pr && atoi(pr) >= 3)
			w->batch_mode |= 2;
		// bits 2, 3: the verdict on a round forced (LRZGPU_RESOLVE_POOR); 4: the dense variant may be asked for; 5: ... and
		// never gives back -- all three read per scan (scan_chunk_device)
	}
	w->seg_cap = (size_t)1 << 28; // up to 256 Mi positions per segment
	if ((int64_t)w->seg_cap > max_chunk + TILE)
		w->seg_cap = (size_t)(((max_chunk + TILE) / TILE + 1) * TILE);
	HIPCHK(hipMalloc(&w->cand_rel, w->seg_cap * 4));
	HIPCHK(hipMalloc(&w->cand_tag, w->seg_cap * 8));
	HIPCHK(hipMalloc(&w->tile_count, (w->seg_cap / TILE + 2) * 4));
	HIPCHK(hipMalloc(&w->tile_base, (w->seg_cap / TILE + 2) * 4));
	w->comp_cap = w->seg_cap < ((size_t)1 << 25) ? w->seg_cap : (size_t)1 << 25; // segments aim at ~2M candidates
	if (const char *e = getenv("LRZGPU_COMP_CAP")) { // test hook: small values force the per-tile fallback
		const long long v = atoll(e);
		if (v >= 0 && (size_t)v < w->comp_cap)
			w->comp_cap = (size_t)v;
	}
	HIPCHK(hipMalloc(&w->comp_rel, (w->comp_cap + 64) * 4));
	HIPCHK(hipMalloc(&w->comp_tag, (w->comp_cap + 64) * 8));
	w->rec_cap = max_chunk / MINIMUM_MATCH + 16;
	if (w->rec_cap > (int64_t)1 << 26)
		w->rec_cap = (int64_t)1 << 26;
	HIPCHK(hipMalloc(&w->records, (size_t)w->rec_cap * sizeof(MatchRec)));
	w->crc_cap = (size_t)(max_chunk / CRC_TILE + 2);
	HIPCHK(hipMalloc(&w->crc_partial, (w->crc_cap + 256) * 4));
	HIPCHK(hipMalloc(&w->long_best, 8));
	uint64_t hx[256];
	hash_index_table(hx);
	HIPCHK(hipMemcpy(w->hx, hx, sizeof(hx), hipMemcpyHostToDevice));
	*out = w;
	return 0;
}

void scan_workspace_destroy(ScanWorkspace *w)
{
	if (!w)
		return;
	void *ptrs[] = {w->table, w->state, w->hx, w->cand_rel, w->cand_tag, w->tile_count, w->records, w->crc_partial, w->long_best,
			w->tile_base, w->comp_rel, w->comp_tag, w->rank_bytes, w->fp_bytes};
	for (void *p : ptrs)
		if (p)
			(void)hipFree(p);
	free(w);
}

int scan_chunk_device(ScanWorkspace *w, const uint8_t *d_chunk, int64_t chunk_size, int rzip_level,
		      int64_t *victim_round, ScanResult *res, hipStream_t s, const ScanProgressFn &progress, bool census)
{
	unsigned mb, freq, chain;
	rzip_level_params(rzip_level, &mb, &freq, &chain);
	ScanState h;
	memset(&h, 0, sizeof(h));
	h.chunk_size = chunk_size;
	h.end = chunk_size - MINIMUM_MATCH;
	h.hash_bits = w->hash_bits;
	h.max_chain_len = chain;
	h.hash_limit = ((int64_t)1 << w->hash_bits) / 3 * 2;
	h.tag_mask = ((uint64_t)1 << freq) - 1;
	h.min_mask = h.tag_mask;
	h.victim_round = *victim_round;
	h.rec_cap = w->rec_cap;
	h.hint_p = h.hint_op = -1;
	HIPCHK(hipMemsetAsync(w->table, 0, ((size_t)16 << w->hash_bits) + 64 * 16, s));
	HIPCHK(hipMemsetAsync(w->rank_bytes, 0, ((size_t)1 << w->hash_bits) + 256, s));
	HIPCHK(hipMemsetAsync(w->fp_bytes, 0, ((size_t)1 << w->hash_bits) + 256, s));
	HIPCHK(hipMemcpyAsync(w->state, &h, sizeof(h), hipMemcpyHostToDevice, s));

	// test hook, read per scan: the resolver's verdict on its rounds forced either way (rzip_resolve_mw.h) -- the rounds and
	// the stretches of exact steps are two implementations of one automaton, and the suite runs every data kind through both
	int batch_mode = w->batch_mode;
	if (const char *e = getenv("LRZGPU_RESOLVE_POOR")) {
		if (!strcmp(e, "never"))
			batch_mode |= 4;
		else if (!strcmp(e, "always"))
			batch_mode |= 8;
	}
	// The resolver has two variants that hand over to each other between launches: k_resolve_mw<4> (four wavefronts, 256
	// candidates a round: a real match or a neighbour's write into a candidate's probe run ends the round) and the dense
	// one (one wavefront: matches are carried through the round by an in-order pass, round-robin evictions of one tag
	// pass each other) for inputs where the first commits two or three candidates a round -- a few symbols, short
	// phrases over and over.  LRZGPU_RESOLVE_DENSE=0: never (exact stretches instead, rounds 1 to 5); =always: every
	// launch is the dense one (tests: every data kind through it)
	bool dense = false;
	// a hand-over that did not pay (the dense variant gave back: seven rounds in eight had no use for it) is not asked for
	// again for 1, 2, 4 ... 64 segments: data whose rounds are poor for other reasons -- a match of 64 KiB after every
	// mutated byte, as in the headline file's later chunks -- would otherwise go back and forth every few hundred rounds,
	// each time with a new K1 pass over the rest of the segment
	int dense_ban = 0, dense_ban_next = 1;
	const int batch_mode_base = [&] {
		const char *e = getenv("LRZGPU_RESOLVE_DENSE");
		if ((!e || strcmp(e, "0")) && chain <= (unsigned)MAX_EQS)
			batch_mode |= 16; // (level 9's chains of 128 equal tags are beyond what a lane of the variant holds: exact stretches there)
		if (e && !strcmp(e, "always")) {
			batch_mode |= 32;
			dense = true;
		}
		return batch_mode;
	}();
	const auto wall0 = std::chrono::steady_clock::now();
	const int64_t end = h.end;
	int64_t p_skip = 0;
	uint64_t min_mask = h.min_mask;
	if (census && end > 0 && ((uintptr_t)d_chunk & 15) == 0) {
		// incompressible data: no 31-byte window occurs twice, so no candidate can become a match -- the automaton is
		// not run (its table, its statistics and the victim_round it would end with are of no use to anyone here)
		int dev = 0;
		(void)hipGetDevice(&dev);
		CensusStats cs;
		EventTimer tc(s);
		const int verdict = duplicate_census(d_chunk, chunk_size, dev, s, &cs);
		tc.stop();
		HIPCHK(stream_wait(s));
		if (verdict < 0)
			return -5;
		if (getenv("LRZGPU_TRACE"))
			fprintf(stderr, "lrzgpu scan: census of %lld bytes in %.1f ms: sample %lld anchors / %lld equal, all %lld anchors / %lld equal (%lld of them chance): %s\n",
				(long long)chunk_size, tc.ms(), (long long)cs.sample_anchors, (long long)cs.sample_equal, (long long)cs.anchors, (long long)cs.equal, (long long)cs.cleared,
				verdict == 1 ? "no 31-byte window occurs twice, the resolver is not run" : "the resolver runs");
		{
			ProfileStore &ps = ProfileStore::get();
			std::lock_guard<std::mutex> lk(ps.mu);
			ps.p.tag_scan_ms += tc.ms_noted(ps, PK_TAG_SCAN); // (booked with the other all-CU streaming kernel of the scan)
			ps.p.tag_scan_launches++;
			ps.p.tag_scan_positions += verdict == 1 ? chunk_size : chunk_size / 64;
		}
		if (verdict == 1) {
			p_skip = end; // every candidate "examined": nothing left for the loop below
			h.p_skip = end;
			HIPCHK(hipMemcpyAsync(w->state, &h, sizeof(h), hipMemcpyHostToDevice, s));
			if (progress) {
				int pr = progress(h, p_skip);
				if (pr)
					return pr;
			}
		}
	}
	while (p_skip + 1 <= end) {
		const int64_t seg_lo = (p_skip + 1) & ~(int64_t)15; // (k_tag_scan reads aligned 16-byte words; the resolver drops pos <= p_skip)
		// size the segment for ~2M candidates under the current mask
		int mbits = __builtin_popcountll(min_mask);
		int64_t seg = (int64_t)(2 << 20) << (mbits > 9 ? 9 : mbits);
		if (seg < (4 << 20)) // small early segments: the back end gets its first blocks sooner
			seg = 4 << 20;
		if (const char *e = getenv("LRZGPU_SEG_BYTES")) { // test hook: fixed segment size
			const long long v = atoll(e);
			if (v >= TILE)
				seg = v;
		}
		if (seg > (int64_t)w->seg_cap - TILE)
			seg = (int64_t)w->seg_cap - TILE;
		int64_t seg_hi = seg_lo + seg;
		if (seg_hi > end + 1)
			seg_hi = end + 1;
		const int ntiles = (int)((seg_hi - seg_lo + TILE - 1) / TILE);
		EventTimer t1(s);
		hipLaunchKernelGGL(k_tag_scan, dim3(ntiles), dim3(256), 0, s, d_chunk, (i64)seg_lo, (i64)seg_hi, (const u64 *)w->hx,
				   (const ScanState *)w->state, w->cand_rel, (u64 *)w->cand_tag, w->tile_count);
		hipLaunchKernelGGL(k_tile_scan, dim3(1), dim3(1024), 0, s, (const uint32_t *)w->tile_count, ntiles, w->tile_base);
		hipLaunchKernelGGL(k_compact_cands, dim3(ntiles), dim3(256), 0, s, (const uint32_t *)w->cand_rel, (const u64 *)w->cand_tag,
				   (const uint32_t *)w->tile_count, (const uint32_t *)w->tile_base, (uint32_t)w->comp_cap, w->comp_rel,
				   (u64 *)w->comp_tag);
		t1.stop();
		EventTimer t2(s);
		if (dense)
			hipLaunchKernelGGL((k_resolve_mw<1, DENSE_HITS, MAX_EQS, true>), dim3(1), dim3(64), 0, s, d_chunk, (Slot *)w->table, w->state, (i64)seg_lo,
					   ntiles, (const uint32_t *)w->cand_rel, (const u64 *)w->cand_tag, (const uint32_t *)w->tile_count, w->records,
					   batch_mode_base, (const uint32_t *)w->tile_base, (const uint32_t *)w->comp_rel, (const u64 *)w->comp_tag,
					   (uint32_t)w->comp_cap, w->rank_bytes, w->fp_bytes);
		else
			hipLaunchKernelGGL((k_resolve_mw<4, MAX_HITS, MAX_EQS>), dim3(1), dim3(256), 0, s, d_chunk, (Slot *)w->table, w->state, (i64)seg_lo,
					   ntiles, (const uint32_t *)w->cand_rel, (const u64 *)w->cand_tag, (const uint32_t *)w->tile_count, w->records,
					   dense_ban > 0 ? (batch_mode_base & ~16) : batch_mode_base, (const uint32_t *)w->tile_base, (const uint32_t *)w->comp_rel,
					   (const u64 *)w->comp_tag, (uint32_t)w->comp_cap, w->rank_bytes, w->fp_bytes);
		t2.stop();
		HIPCHK(d2h_pageable(&h, w->state, sizeof(h), s)); // (sleeps while the resolver runs)
		if (getenv("LRZGPU_TRACE"))
			fprintf(stderr, "lrzgpu scan: seg [%lld,%lld) tiles %d  k1 %.2f ms  k2 %.2f ms  p_skip %lld  mask %llx  lookups %lld recs %lld  batches %lld committed %lld serial %lld complex %lld conflict %lld\n",
				(long long)seg_lo, (long long)seg_hi, ntiles, t1.ms(), t2.ms(), (long long)h.p_skip,
				(unsigned long long)h.min_mask, (long long)h.lookups, (long long)h.n_records, (long long)h.dbg[0],
				(long long)h.dbg[1], (long long)h.dbg[2], (long long)h.dbg[3], (long long)h.dbg[5]);
		{
			ProfileStore &ps = ProfileStore::get();
			std::lock_guard<std::mutex> lk(ps.mu);
			ps.p.tag_scan_ms += t1.ms_noted(ps, PK_TAG_SCAN);
			ps.p.tag_scan_launches++;
			ps.p.tag_scan_positions += seg_hi - seg_lo;
			ps.p.resolve_ms += t2.ms_noted(ps, PK_RESOLVE);
			ps.p.resolve_launches++;
		}
		if (h.error == 3) {
			// the resolver met a match that is still equal after LONG_EXTENT bytes: finish the
			// compare on the whole GPU, leave the answer as a hint and resume at that candidate
			const int64_t limit = chunk_size - h.ext_p;
			unsigned long long best = (unsigned long long)limit;
			HIPCHK(hipMemcpyAsync(w->long_best, &best, 8, hipMemcpyHostToDevice, s));
			EventTimer t3(s);
			hipLaunchKernelGGL(k_long_compare, dim3(2048), dim3(256), 0, s, d_chunk, (i64)h.ext_p, (i64)h.ext_op, (i64)h.ext_done,
					   (i64)limit, w->long_best);
			t3.stop();
			HIPCHK(d2h_pageable(&best, w->long_best, 8, s));
			h.hint_p = h.ext_p;
			h.hint_op = h.ext_op;
			h.hint_len = (int64_t)best;
			{
				ProfileStore &ps = ProfileStore::get();
				std::lock_guard<std::mutex> lk(ps.mu);
				ps.p.long_compare_ms += t3.ms_noted(ps, PK_LONG_COMPARE);
				ps.p.long_compare_launches++;
				ps.p.long_compare_bytes += 2 * ((int64_t)best - h.ext_done);
			}
			h.error = 0;
			HIPCHK(hipMemcpyAsync(w->state, &h, sizeof(h), hipMemcpyHostToDevice, s));
			if (getenv("LRZGPU_TRACE"))
				fprintf(stderr, "lrzgpu scan: long extent at %lld from %lld: %lld bytes, k3 %.2f ms\n", (long long)h.ext_p,
					(long long)h.ext_op, (long long)best, t3.ms());
			p_skip = h.p_skip;
			min_mask = h.min_mask;
			if (progress) {
				int pr = progress(h, p_skip);
				if (pr)
					return pr;
			}
			continue;
		}
		if (h.error == 4 || h.error == 5) {
			// the variants hand over: the other one goes on at the first candidate this one has not examined
			if (getenv("LRZGPU_TRACE"))
				fprintf(stderr, "lrzgpu scan: the %s resolver takes over at %lld\n", h.error == 4 ? "dense" : "four-wavefront", (long long)h.p_skip + 1);
			dense = h.error == 4;
			if (h.error == 5) {
				dense_ban = dense_ban_next;
				dense_ban_next = dense_ban_next < 64 ? 2 * dense_ban_next : 64;
			}
			h.error = 0;
			HIPCHK(hipMemcpyAsync(w->state, &h, sizeof(h), hipMemcpyHostToDevice, s));
			p_skip = h.p_skip;
			min_mask = h.min_mask;
			if (progress) {
				int pr = progress(h, p_skip);
				if (pr)
					return pr;
			}
			continue;
		}
		if (h.error)
			return h.error == 1 ? -4 : -5;
		if (dense)
			dense_ban_next = 1; // (the variant saw a segment out: it is where it belongs)
		else if (dense_ban > 0)
			dense_ban--;
		p_skip = h.p_skip > seg_hi - 1 ? h.p_skip : seg_hi - 1;
		min_mask = h.min_mask;
		if (progress) {
			int pr = progress(h, p_skip);
			if (pr)
				return pr;
		}
	}
	if (chunk_size > 0 && end > 0) {
		// state already in h from the last segment
	} else {
		HIPCHK(d2h_pageable(&h, w->state, sizeof(h), s));
	}
	res->records.resize((size_t)h.n_records);
	if (h.n_records)
		HIPCHK(d2h_pageable(res->records.data(), w->records, (size_t)h.n_records * sizeof(MatchRec), s));
	uint32_t crc = 0;
	if (crc32_device(w, d_chunk, chunk_size, &crc, s) != 0)
		return -6;
	HIPCHK(stream_wait(s));
	res->crc = crc;
	res->final_state = h;
	if (getenv("LRZGPU_TRACE")) {
		fprintf(stderr, "lrzgpu scan: rounds %lld committed %lld serial steps %lld  hash_count %lld\n", (long long)h.dbg[0], (long long)h.dbg[1],
			(long long)h.dbg[2], (long long)h.hash_count);
	}
	*victim_round = h.victim_round;
	{
		int64_t mb = 0;
		for (const MatchRec &r : res->records)
			mb += r.len;
		ProfileStore &ps = ProfileStore::get();
		std::lock_guard<std::mutex> lk(ps.mu);
		ps.p.resolve_lookups += h.lookups;
		ps.p.resolve_inserts += h.inserts;
		ps.p.resolve_match_bytes += mb;
		for (int k = 0; k < 16; k++)
			ps.p.resolve_dbg[k] += h.dbg[k];
		ps.p.scan_wall_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - wall0).count();
	}
	return 0;
}

} // namespace lrzgpu
