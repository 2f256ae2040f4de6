// lzma_mf.hip -- GPU match finder for the per-block LZMA backend (gfx950 / MI355X).
//
// Produces, for every position of a block, exactly the (len, dist-1) list the reference's
// multithreaded BT4 finder hands to the parser (reference src/lzma/C/LzFindMt.c:1274-1317 after
// MixMatches3 1031-1072; tree walk LzFindOpt.c:67-244; heads LzFindMt.c:368-394).
//
// MI355X-first decomposition (the reference walks positions strictly in order on one thread):
//   * a binary tree only ever links positions that share one main-hash value, and the hash-head
//     chain of the reference is "previous position with the same hash value".  So the block is
//     partitioned by hash value with a stable radix sort (key = hash, value = position) and every
//     hash bucket becomes an independent serial chain -> one GPU lane per bucket, buckets
//     scheduled longest-first.  Tree nodes live in SORTED order (a bucket's nodes are contiguous), 32 bytes
//     each: two links, the position and the first 20 bytes at it -- one aligned load per visit (see BtNode)
//     -- instead of the reference's cyclic son[] pairs; the `delta >= cyclicBufferSize` cut-off is kept.
//   * the h2/h3 "most recent position with the same 2/3-byte hash" tables of the LZ thread are
//     pure functions of the data (updated at every position), obtained with two more sorts.
//   * records are written into a bump-allocated pool and then gathered into position order so the
//     host parser streams them sequentially.
//   * runs of one byte value put millions of consecutive positions into ONE bucket; each of them
//     repeats its predecessor's result (full-length match at distance 1, inherited sons), so the
//     lane that owns such a bucket writes them eight at a time without reading the tree.
// Levels 1-4 (algo 0) use hash chains instead (HC5, LzFind.c:880-958, 1431-1502): every chain link
// is "previous position with the same 5-byte hash" and no position changes what a later one sees,
// so k_hc5 is one thread per position (see the comment above it).
// Roofline: latency bound pointer chasing (the longest bucket's serial walk sets the launch time), no MFMA.
// Algorithmic bytes/position: 1 B read + 4 B head r/w + 8 B son pair (+ <=48 node visits) -- see DESIGN.md.
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/device/device_select.hpp>
#include <rocprim/iterator/counting_iterator.hpp>
#include <rocprim/iterator/transform_iterator.hpp>

#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "common.h"
#include "lzma_mf.h"
#include "pools.h"
#include "profile.h"

namespace lrzgpu {

#define HIPCHK(x)                                                                              \
	do {                                                                                   \
		hipError_t e_ = (x);                                                           \
		if (e_ != hipSuccess) {                                                        \
			fprintf(stderr, "lrzgpu: HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
			return -1;                                                             \
		}                                                                              \
	} while (0)

__device__ __forceinline__ uint32_t crc_byte(uint32_t b)
{
	uint32_t r = b;
#pragma unroll
	for (int j = 0; j < 8; j++)
		r = (r >> 1) ^ (0xEDB88320u & (0u - (r & 1)));
	return r;
}

// keys for the hash tables; which: 4 = BT4 main hash (GetHeads4 / GetHeads4b), 5 = HC5 main hash, 3 = h3, 2 = h2
template <int WHICH>
__global__ void __launch_bounds__(256) k_keys(const uint8_t *__restrict__ src, uint32_t n4, uint32_t mask, int big,
					      uint32_t *__restrict__ keys, uint32_t *__restrict__ vals)
{
	uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	uint32_t stride = gridDim.x * blockDim.x;
	for (; i < n4; i += stride) {
		uint32_t b0 = src[i], b1 = src[i + 1], b2 = src[i + 2], b3 = src[i + 3];
		uint32_t c0 = crc_byte(b0);
		uint32_t k;
		if (WHICH == 4) {
			if (big)
				k = (c0 & mask) ^ (b1 | (b2 << 8) | (b3 << 16));
			else
				k = (c0 & mask) ^ ((crc_byte(b3) << 5) & mask) ^ (b1 | (b2 << 8));
		} else if (WHICH == 5) { // HASH5_CALC, LzFind.c:56-63 (positions with >= 5 bytes only)
			k = ((c0 ^ b1) ^ (b2 << 8) ^ (crc_byte(b3) << 5) ^ (crc_byte(src[i + 4]) << 10)) & mask;
		} else if (WHICH == 3) {
			k = ((c0 ^ b1) ^ (b2 << 8)) & 0xFFFF;
		} else {
			k = (c0 ^ b1) & 1023;
		}
		keys[i] = k;
		vals[i] = i;
	}
}

// prev[pos] = 1-based position of the previous element with the same key, 0 if none
__global__ void __launch_bounds__(256) k_link_prev(const uint32_t *__restrict__ skey, const uint32_t *__restrict__ sval,
						   uint32_t n4, uint32_t *__restrict__ prev)
{
	uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
	uint32_t stride = gridDim.x * blockDim.x;
	for (; k < n4; k += stride) {
		uint32_t p = 0;
		if (k > 0 && skey[k] == skey[k - 1])
			p = sval[k - 1] + 1;
		prev[sval[k]] = p;
	}
}

__global__ void __launch_bounds__(256) k_flag_heads(const uint32_t *__restrict__ skey, uint32_t n4, uint8_t *__restrict__ flags)
{
	uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
	uint32_t stride = gridDim.x * blockDim.x;
	for (; k < n4; k += stride)
		flags[k] = (k == 0 || skey[k] != skey[k - 1]) ? 1 : 0;
}

// bucket lengths; n_long[0] = how many of them have at least long_min positions, n_long[1] = at least mid_min (the
// buckets are sorted by length afterwards, so those counts are the boundaries between the walk kernels: k_bt_group<64>,
// k_bt_group<8>, k_bt)
__global__ void __launch_bounds__(256) k_seg_len(const uint32_t *__restrict__ seg_start, const uint32_t *__restrict__ nseg_p,
						 uint32_t n4, uint32_t *__restrict__ seg_len, uint32_t long_min, uint32_t mid_min,
						 uint32_t *__restrict__ n_long)
{
	uint32_t nseg = *nseg_p;
	uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
	uint32_t stride = gridDim.x * blockDim.x;
	uint32_t mine = 0, mid = 0;
	for (; s < nseg; s += stride) {
		const uint32_t len = (s + 1 < nseg ? seg_start[s + 1] : n4) - seg_start[s];
		seg_len[s] = len;
		mine += len >= long_min ? 1u : 0u;
		mid += len >= mid_min ? 1u : 0u;
	}
	if (mine)
		atomicAdd(n_long, mine);
	if (mid)
		atomicAdd(n_long + 1, mid);
}

constexpr int kMaxRec = 128; // u32 entries per position: 2 * (2 hash pairs + cut (<= 48)) -> 100

// One lane per hash bucket: replay the bucket's positions in order through the BT4 tree walk
// (LzFindOpt.c GetMatchesSpecN_2 semantics), then the LZ-thread merge (MixMatches3).
//
// Tree storage.  The reference keeps son[2 * cyclic position]; a tree only ever links positions of ONE bucket, but
// those are a dictionary apart in position order, so indexing by position makes every node visit two random
// sectors (the son pair, and the bytes at that position to compare with) -- ~20 visits per position on text.
// Here the nodes live in SORTED order: node s belongs to the s-th entry of the hash-sorted position list, so a
// bucket's nodes are contiguous (most buckets fit a few cache lines, long ones stay L2-resident while their lane
// walks them), links are sorted indices (+1, 0 = none), and each 32-byte node carries its position and the
// first 20 bytes at that position: a visit is ONE aligned 32-byte load; the block itself is only read for the
// current position and for agreements beyond 20 bytes.
struct __attribute__((aligned(32))) BtNode {
	uint32_t son0, son1; // the reference's pair[0] / pair[1], as sorted index + 1
	uint32_t pos;        // 1-based position (pos of the reference)
	uint32_t w[5];       // bytes 0..19 at the position, little endian
};
static_assert(sizeof(BtNode) == 32, "one sector per node");
struct __attribute__((packed, aligned(1))) PackedU64 {
	uint64_t v;
};
struct __attribute__((packed, aligned(1))) PackedU32 {
	uint32_t v;
};
__device__ __forceinline__ uint64_t load_u64(const uint8_t *p) { return reinterpret_cast<const PackedU64 *>(p)->v; }
__device__ __forceinline__ uint32_t load_u32(const uint8_t *p) { return reinterpret_cast<const PackedU32 *>(p)->v; }
constexpr uint32_t kNodeBytes = 20;
// first index in [0, 20) where two 20-byte prefixes differ (20 if none)
__device__ __forceinline__ uint32_t prefix_mismatch(const uint32_t a[5], const uint32_t b[5])
{
	const uint64_t x0 = ((uint64_t)(a[1] ^ b[1]) << 32) | (a[0] ^ b[0]);
	if (x0)
		return (uint32_t)(__ffsll((long long)x0) - 1) >> 3;
	const uint64_t x1 = ((uint64_t)(a[3] ^ b[3]) << 32) | (a[2] ^ b[2]);
	if (x1)
		return 8 + ((uint32_t)(__ffsll((long long)x1) - 1) >> 3);
	const uint32_t x2 = a[4] ^ b[4];
	return x2 ? 16 + ((uint32_t)(__ffs((int)x2) - 1) >> 3) : 20;
}
// (selects, not a[k >> 2]: a register array indexed by a variable goes to scratch memory)
__device__ __forceinline__ uint32_t prefix_byte(const uint32_t a[5], uint32_t k)
{
	const uint32_t w = k < 4 ? a[0] : k < 8 ? a[1] : k < 12 ? a[2] : k < 16 ? a[3] : a[4];
	return (w >> (8 * (k & 3))) & 0xFF;
}

// bytes 0..19 at a position as a node prefix (zero beyond the end of the block: never compared)
__device__ __forceinline__ void node_prefix(const uint8_t *cur, uint32_t avail, uint32_t w[5])
{
	if (avail >= kNodeBytes) {
		const uint64_t q0 = load_u64(cur), q1 = load_u64(cur + 8);
		w[0] = (uint32_t)q0;
		w[1] = (uint32_t)(q0 >> 32);
		w[2] = (uint32_t)q1;
		w[3] = (uint32_t)(q1 >> 32);
		w[4] = load_u32(cur + 16);
	} else {
#pragma unroll
		for (int k = 0; k < 5; k++)
			w[k] = 0;
#pragma unroll
		for (uint32_t k = 0; k < kNodeBytes; k++) // (static indices)
			if (k < avail)
				w[k >> 2] |= (uint32_t)cur[k] << (8 * (k & 3));
	}
}

// agreement of the current position with a visited node: starts from min(len0, len1) like the reference, the first 20
// bytes from the node itself, the rest from the block (unaligned loads stay inside [0, len_limit) <= avail)
__device__ __forceinline__ uint32_t agree_len(const uint32_t nw[5], const uint32_t mw[5], const uint8_t *cur, const uint8_t *pb, uint32_t len,
					      uint32_t len_limit)
{
	if (len < kNodeBytes) {
		const uint32_t m = prefix_mismatch(nw, mw);
		const uint32_t lim = len_limit < kNodeBytes ? len_limit : kNodeBytes;
		len = m < lim ? m : lim;
	}
	if (len >= kNodeBytes && len < len_limit) {
		while (len + 32 <= len_limit) {
			uint64_t x[4];
#pragma unroll
			for (int w = 0; w < 4; w++)
				x[w] = load_u64(pb + len + 8 * w) ^ load_u64(cur + len + 8 * w);
#pragma unroll
			for (int w = 0; w < 4; w++)
				if (x[w])
					return len + 8 * w + ((uint32_t)(__ffsll((long long)x[w]) - 1) >> 3);
			len += 32;
		}
		while (len + 8 <= len_limit) {
			const uint64_t x = load_u64(pb + len) ^ load_u64(cur + len);
			if (x)
				return len + ((uint32_t)(__ffsll((long long)x) - 1) >> 3);
			len += 8;
		}
		while (len != len_limit && pb[len] == cur[len])
			++len;
	}
	return len;
}

// LZ-thread merge: MixMatches3 (h2/h3 candidates nearer than the first tree match); returns the number of entries
__device__ __forceinline__ uint32_t mix_matches(const uint8_t *__restrict__ src, const uint8_t *cur, uint32_t i, uint32_t pos, uint32_t dict,
						 bool have_tree, uint32_t first_dist1, const uint32_t *__restrict__ prev2,
						 const uint32_t *__restrict__ prev3, uint32_t mix[4])
{
	uint32_t nmix = 0;
	const uint32_t min_pos = have_tree ? pos - first_dist1 : (pos > dict ? pos - dict : 1);
	// (a first tree match at distance 1 leaves nothing nearer: no loads at all)
	const uint32_t c2 = min_pos < pos ? prev2[i] : 0, c3 = min_pos < pos ? prev3[i] : 0;
	bool done = false;
	if (c2 >= min_pos && src[c2 - 1] == cur[0]) {
		mix[1] = pos - c2 - 1;
		if (src[c2 - 1 + 2] == cur[2]) {
			mix[0] = 3;
			done = true;
		} else
			mix[0] = 2;
		nmix = 2;
	}
	if (!done && c3 >= min_pos && src[c3 - 1] == cur[0]) {
		if (nmix == 0) { // (static indices: mix[] stays in registers)
			mix[0] = 3;
			mix[1] = pos - c3 - 1;
		} else {
			mix[2] = 3;
			mix[3] = pos - c3 - 1;
		}
		nmix += 2;
	}
	return nmix;
}
__device__ __forceinline__ void put_mix(uint32_t *o, const uint32_t mix[4], uint32_t nmix)
{
	if (nmix >= 2) {
		o[0] = mix[0];
		o[1] = mix[1];
	}
	if (nmix == 4) {
		o[2] = mix[2];
		o[3] = mix[3];
	}
}

// Everything about a position that does not depend on the tree: where it is, its first 20 bytes, its h2 / h3
// candidates and the bytes MixMatches3 looks at there.  Loaded AHEAD of the position's walk (the lane kernel keeps a
// three-stage pipeline of these, the wave kernel stages 64 at a time in LDS), so that the per-position chain of
// dependent loads is the tree walk alone.
struct PosData {
	uint32_t i;
	uint32_t w[5];
	uint32_t c2, c3;
	uint32_t bytes; // src[c2 - 1] | src[c2 + 1] << 8 | src[c3 - 1] << 16 (0 where there is no candidate)
};
__device__ __forceinline__ uint32_t cand_bytes(const uint8_t *__restrict__ src, uint32_t c2, uint32_t c3)
{
	uint32_t b = 0;
	if (c2)
		b = (uint32_t)src[c2 - 1] | ((uint32_t)src[c2 + 1] << 8);
	if (c3)
		b |= (uint32_t)src[c3 - 1] << 16;
	return b;
}
__device__ __forceinline__ void load_pos(PosData &d, const uint8_t *__restrict__ src, uint32_t n, const uint32_t *__restrict__ spos, uint32_t sorted_index,
					 const uint32_t *__restrict__ prev2, const uint32_t *__restrict__ prev3)
{
	d.i = spos[sorted_index];
	node_prefix(src + d.i, n - d.i, d.w);
	d.c2 = prev2[d.i];
	d.c3 = prev3[d.i];
	d.bytes = cand_bytes(src, d.c2, d.c3);
}
// MixMatches3 from preloaded candidates (the position has at least 4 bytes: it is in a bucket)
__device__ __forceinline__ uint32_t mix_from(const PosData &d, uint32_t pos, uint32_t dict, bool have_tree, uint32_t first_dist1, uint32_t mix[4])
{
	uint32_t nmix = 0;
	const uint32_t min_pos = have_tree ? pos - first_dist1 : (pos > dict ? pos - dict : 1);
	const uint32_t cur0 = d.w[0] & 0xFF, cur2 = (d.w[0] >> 16) & 0xFF;
	bool done = false;
	if (d.c2 >= min_pos && (d.bytes & 0xFF) == cur0) {
		mix[1] = pos - d.c2 - 1;
		if (((d.bytes >> 8) & 0xFF) == cur2) {
			mix[0] = 3;
			done = true;
		} else
			mix[0] = 2;
		nmix = 2;
	}
	if (!done && d.c3 >= min_pos && ((d.bytes >> 16) & 0xFF) == cur0) {
		if (nmix == 0) {
			mix[0] = 3;
			mix[1] = pos - d.c3 - 1;
		} else {
			mix[2] = 3;
			mix[3] = pos - d.c3 - 1;
		}
		nmix += 2;
	}
	return nmix;
}

// Output space for the lists of one wavefront: the wave takes the pool in pieces (one returning global atomic per
// kChunk entries instead of one per position -- a ~1 us round trip each) and hands them out with a prefix sum over the
// lanes' counts.  Every lane of the wave must call this (cnt = 0 for lanes with nothing to write).
struct WaveAlloc {
	unsigned long long base; // wave-uniform
	uint32_t free_;
	uint32_t chunk; // entries per piece: 8192 on real blocks, less where the pool itself is small
};
// every wave may leave one piece partly unused: all of them together get at most half the pool
static inline uint32_t pool_chunk(unsigned long long pool_cap, unsigned long long waves)
{
	const unsigned long long c = pool_cap / 2 / (waves ? waves : 1);
	return (uint32_t)(c < 16 ? 16 : (c > 8192 ? 8192 : c));
}
__device__ __forceinline__ unsigned long long wave_take(WaveAlloc &a, uint32_t cnt, unsigned long long *__restrict__ cursor)
{
	// inclusive prefix sum over the 64 lanes
	uint32_t incl = cnt;
#pragma unroll
	for (int o = 1; o < 64; o <<= 1) {
		const uint32_t v = __shfl_up(incl, o);
		if ((int)(threadIdx.x & 63) >= o)
			incl += v;
	}
	const uint32_t total = __shfl(incl, 63);
	if (total == 0)
		return 0;
	if (total > a.free_) { // (what is left of the old piece is simply not used)
		const uint32_t take = total > a.chunk ? total : a.chunk;
		unsigned long long b = 0;
		if ((threadIdx.x & 63) == 0)
			b = atomicAdd(cursor, (unsigned long long)take);
		a.base = __shfl(b, 0);
		a.free_ = take;
	}
	const unsigned long long st = a.base + (incl - cnt);
	a.base += total;
	a.free_ -= total;
	return st;
}

// Match records of the walk in progress, one column per lane, in dynamic LDS sized for the launch's `cut`: a walk visits
// at most `cut` nodes and every visit adds at most one (length, distance) record, so cut pairs per lane are enough --
// 4-byte distances and 2-byte lengths (<= 273): 18 KB per wavefront at the cut of 48 of levels 7-9 instead of the 32 KB of
// a fixed 128-entry column, i.e. seven wavefronts per CU where LDS allowed four (k_bt_group) or five (k_bt).
struct RecColumns {
	uint32_t *dist; // [pair][64]
	uint16_t *len;  // [pair][64]
	__device__ __forceinline__ void put(uint32_t pair, uint32_t lane, uint32_t l, uint32_t d) const
	{
		dist[pair * 64 + lane] = d;
		len[pair * 64 + lane] = (uint16_t)l;
	}
	// entry k of the list as the finder writes it out: even = length, odd = distance - 1
	__device__ __forceinline__ uint32_t entry(uint32_t k, uint32_t lane) const { return (k & 1) ? dist[(k >> 1) * 64 + lane] : len[(k >> 1) * 64 + lane]; }
};
static inline size_t rec_lds_bytes(uint32_t cut) { return (size_t)(cut < 1 ? 1 : cut) * 64 * 6; }
__device__ __forceinline__ RecColumns rec_columns(uint32_t cut)
{
	extern __shared__ __attribute__((aligned(16))) uint8_t rec_lds_raw[];
	const uint32_t pairs = cut < 1 ? 1 : cut;
	RecColumns r;
	r.dist = reinterpret_cast<uint32_t *>(rec_lds_raw);
	r.len = reinterpret_cast<uint16_t *>(rec_lds_raw + (size_t)pairs * 64 * 4);
	return r;
}

// ---- short buckets: one lane per bucket ---------------------------------------------------------------------------
// The lane replays its bucket's positions in order.  Match records are collected in LDS (one column per lane), output
// space comes from wave_take().  The loop over the bucket is wave-uniform (lanes whose bucket is done idle along), so
// that the allocation can be a wave operation; buckets are scheduled by length, a wave's 64 buckets are alike.
__device__ __forceinline__ void bt_lane_body(const uint32_t block, const uint8_t *__restrict__ src, uint32_t n,
					     const uint32_t *__restrict__ spos,
					     const uint32_t *__restrict__ seg_len_sorted, const uint32_t *__restrict__ seg_start_sorted,
					     const uint32_t nseg, uint32_t first_seg,
					     BtNode *__restrict__ node,
					     const uint32_t *__restrict__ prev2, const uint32_t *__restrict__ prev3,
					     uint32_t dict, uint32_t fb, uint32_t cut,
					     uint8_t *__restrict__ counts, uint64_t *__restrict__ tmp_start,
					     uint32_t *__restrict__ pool, unsigned long long *__restrict__ cursor,
					     unsigned long long pool_cap, uint32_t chunk, int *__restrict__ err)
{
	const RecColumns rec = rec_columns(cut);
	const uint32_t lane = threadIdx.x;
	const uint32_t g = first_seg + block * 64 + threadIdx.x;
	const bool have = g < nseg;
	const uint32_t k0 = have ? seg_start_sorted[g] : 0;
	const uint32_t L = have ? seg_len_sorted[g] : 0;
	const uint32_t cyc_size = dict + 1;
	uint32_t prev = 0; // 1-based position of the previous element of this bucket
	// runs of one byte value put millions of consecutive positions into one bucket; each of them
	// matches its predecessor over the full fb bytes at the first tree step and inherits its two
	// sons.  Once that has happened for pos-1, pos only has to look at one new byte.
	bool run_ok = false;
	uint32_t run_s0 = 0, run_s1 = 0;
	WaveAlloc wa{0, 0, chunk};
	// the load pipeline: while position j is walked, the bytes MixMatches3 needs of j + 1, the prefix and the h2 / h3
	// candidates of j + 2 and the place of j + 3 are on their way (*_j = the bucket index a stage holds, ~0 = nothing)
	PosData d3{}, d2{};
	uint32_t d3_j = ~0u, d2_j = ~0u, d1_j = ~0u, d1_i = 0;

	for (uint32_t j = 0; __any(j < L); j++) {
		const bool act = j < L;
		uint32_t nrec = 0, nmix = 0, i = 0, pos = 0, len_limit = 0;
		uint32_t mix[4];
		const uint8_t *cur = src;
		PosData D{};
		if (act) {
			const uint32_t self = k0 + j; // sorted index of this position = its node
			if (d3_j == j)
				D = d3;
			else
				load_pos(D, src, n, spos, self, prev2, prev3); // start of the bucket, or behind a run: nothing was on its way
			// move the pipeline on
			if (d2_j == j + 1) {
				d3 = d2;
				d3.bytes = cand_bytes(src, d2.c2, d2.c3);
				d3_j = j + 1;
			} else
				d3_j = ~0u;
			if (d1_j == j + 2) {
				d2.i = d1_i;
				node_prefix(src + d1_i, n - d1_i, d2.w);
				d2.c2 = prev2[d1_i];
				d2.c3 = prev3[d1_i];
				d2_j = j + 2;
			} else
				d2_j = ~0u;
			if (j + 3 < L) {
				d1_i = spos[self + 3];
				d1_j = j + 3;
			} else
				d1_j = ~0u;
			i = D.i;
			pos = i + 1;
			cur = src + i;
			const uint32_t avail = n - i;
			len_limit = avail < fb ? avail : fb;
			const uint32_t cbs = pos < cyc_size ? pos : cyc_size;
			uint32_t delta = pos - prev; // prev == 0 -> delta == pos >= cbs -> empty
			BtNode me;
			me.pos = pos;
#pragma unroll
			for (int k = 0; k < 5; k++)
				me.w[k] = D.w[k];
			me.son0 = me.son1 = 0;

			if (delta >= cbs) {
				node[self] = me;
				run_ok = false;
			} else if (run_ok && delta == 1 && len_limit == fb && cur[fb - 1] == cur[fb - 2]) {
				me.son0 = run_s0;
				me.son1 = run_s1;
				node[self] = me;
				rec.put(0, lane, fb, 0);
				nrec = 2;
			} else {
				run_ok = false;
				node[self] = me;
				// ptr0 / ptr1 of the reference: where the next "greater" / "smaller" subtree root goes.  They
				// start at this position's own pair, then move into visited nodes.
				uint32_t *ptr0 = &node[self].son1, *ptr1 = &node[self].son0;
				uint32_t len0 = 0, len1 = 0, max_len = 3, cv = cut;
				uint32_t cur_ref = self; // sorted index + 1 of the predecessor in the bucket (j > 0 here)
				for (;;) {
					BtNode *np = node + (cur_ref - 1);
					const BtNode N = *np;
					delta = pos - N.pos;
					if (delta >= cbs) {
						*ptr0 = *ptr1 = 0;
						break;
					}
					const uint8_t *pb = cur - delta;
					const uint32_t len = agree_len(N.w, me.w, cur, pb, len0 < len1 ? len0 : len1, len_limit);
					if (max_len < len) {
						max_len = len;
						rec.put(nrec >> 1, lane, len, delta - 1);
						nrec += 2;
						if (len == len_limit) {
							*ptr1 = N.son0;
							*ptr0 = N.son1;
							if (delta == 1 && nrec == 2 && len_limit == fb && pos == prev + 1) {
								run_ok = true; // first step, full length, distance 1
								run_s0 = N.son0;
								run_s1 = N.son1;
							}
							break;
						}
					}
					// (len < len_limit here: a full-length agreement that is not a new maximum cannot happen --
					// max_len < len_limit until the first one, which breaks)
					uint32_t b_node, b_cur; // the bytes that decide the branch
					if (len < kNodeBytes) {
						b_node = prefix_byte(N.w, len);
						b_cur = prefix_byte(me.w, len);
					} else {
						b_node = pb[len];
						b_cur = cur[len];
					}
					uint32_t next;
					if (b_node < b_cur) {
						*ptr1 = cur_ref;
						ptr1 = &np->son1;
						len1 = len;
						next = N.son1;
					} else {
						*ptr0 = cur_ref;
						ptr0 = &np->son0;
						len0 = len;
						next = N.son0;
					}
					if (next >= cur_ref) { // corrupt tree (cannot happen): stop like the reference
						*err = 2;
						*ptr0 = *ptr1 = 0;
						break;
					}
					if (--cv == 0 || next == 0) {
						*ptr0 = *ptr1 = 0;
						break;
					}
					cur_ref = next;
				}
			}
			prev = pos;
			nmix = mix_from(D, pos, dict, nrec != 0, nrec ? rec.entry(1, lane) : 0, mix);
		}
		const uint32_t cnt = nmix + nrec;
		const unsigned long long st = wave_take(wa, cnt, cursor);
		if (act) {
			counts[i] = (uint8_t)cnt;
			if (cnt) {
				tmp_start[i] = st;
				if (st + cnt > pool_cap) {
					*err = 1;
				} else {
					uint32_t *o = pool + st;
					put_mix(o, mix, nmix);
					for (uint32_t k = 0; k < nrec; k++)
						o[nmix + k] = rec.entry(k, lane);
				}
			}
		}

		// Inside a run of one byte value every further position repeats this one: a full-length match
		// at distance 1, no nearer h2/h3 candidate, the predecessor's two sons.  They need no tree
		// reads, so the lane writes them eight at a time (one load of the next bucket entries, one of the
		// next bytes) instead of walking.  (Long run buckets go to k_bt_wave, which does 64 at a time.)
		if (act && run_ok && L >= 64 && len_limit == fb) {
			const uint32_t b = cur[fb - 1];
			uint32_t t = 0; // positions done beyond this one
			unsigned long long loc_base = 0;
			uint32_t loc_free = 0;
			for (;;) {
				if (j + 1 + t + 8 > L || (unsigned long long)i + t + 8 + fb > n)
					break;
				uint32_t sp[8];
#pragma unroll
				for (int q = 0; q < 8; q++)
					sp[q] = spos[k0 + j + 1 + t + q];
				const uint64_t by = load_u64(src + i + t + fb); // byte that position i+t+1+q adds to the window
				uint32_t ok = 0;
#pragma unroll
				for (int q = 0; q < 8; q++)
					if (ok == (uint32_t)q && sp[q] == i + t + 1 + q && ((by >> (8 * q)) & 0xFF) == b)
						ok++;
				for (uint32_t q = 0; q < ok; q++) {
					const uint32_t iq = i + t + 1 + q;
					BtNode rn;
					rn.son0 = run_s0;
					rn.son1 = run_s1;
					rn.pos = iq + 1;
#pragma unroll
					for (int k = 0; k < 5; k++)
						rn.w[k] = b * 0x01010101u; // fb >= 20 bytes of the run lie ahead of every one of them
					node[k0 + j + 1 + t + q] = rn;
					counts[iq] = 2;
					if (loc_free < 2) {
						loc_free = chunk < 1024 ? chunk : 1024;
						loc_base = atomicAdd(cursor, (unsigned long long)loc_free);
					}
					const unsigned long long rs = loc_base;
					loc_base += 2;
					loc_free -= 2;
					tmp_start[iq] = rs;
					if (rs + 2 > pool_cap)
						*err = 1;
					else {
						pool[rs] = fb;
						pool[rs + 1] = 0;
					}
				}
				t += ok;
				if (ok < 8)
					break;
			}
			j += t;
			prev = pos + t;
		}
	}
}

// ---- long buckets: one WAVEFRONT per bucket, walks pipelined ----------------------------------------------------------
// The serial order of a bucket is a chain: the walk of position j starts at the node of position j - 1 and re-roots the
// tree.  But at any moment an unfinished walk owns exactly TWO slots of the tree -- the two link words its next
// "smaller" / "greater" subtree root will be written to (first the two sons of its own node, then sons of visited
// nodes) -- every other word it wrote is final, and a node's position and bytes never change.  The subtree an
// unfinished walk may still visit hangs below those two slots and is reachable from newer roots only THROUGH them.
// So a later walk may run ahead as long as it never reads a slot an earlier unfinished walk still owns: it then sees
// exactly what the serial order would have shown it.
//
// Lane t of the wave runs one walk; walks are started in bucket order as lanes fall free (64 in flight).  A slot that
// is owned is MARKED in memory: the owner stores kPending into it when it takes it (its own node is created with two
// pending sons; moving into a visited node's son marks that son in the same step that resolves the slot left behind)
// and the real value when it resolves it.  A walk that loads kPending where it wants to go stalls and looks again in
// the next round; the oldest unfinished walk never meets a mark (marks only lie in regions no older walk can reach),
// so the wave always makes progress.  All tree words are read and written with (workgroup-scope) atomics and a fence
// separates a round's stores from the next round's loads.
// Runs of one byte value (every position: full-length match with its predecessor at the first step) would serialise
// the pipeline; when the oldest walk in flight hits one the wave switches to writing such positions 64 at a time.
constexpr uint32_t kPending = 0xFFFFFFFFu;
// A bucket's tree is touched by ONE wavefront while k_bt_wave runs: workgroup scope is all the coherence it needs (one
// CU, one vector L1; nothing to write back or invalidate).  Agent scope would be 30 times slower here: on this
// multi-XCD part an agent-scope fence writes the L2 back and invalidates it -- measured 27 us per round.
__device__ __forceinline__ uint32_t ld_coh(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ uint64_t ld_coh64(const uint64_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void st_coh(uint32_t *p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void st_coh64(uint64_t *p, uint64_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void store_node_coh(BtNode *np, const BtNode &v)
{
	uint64_t *q = reinterpret_cast<uint64_t *>(np);
	st_coh64(q + 1, (uint64_t)v.pos | ((uint64_t)v.w[0] << 32));
	st_coh64(q + 2, (uint64_t)v.w[1] | ((uint64_t)v.w[2] << 32));
	st_coh64(q + 3, (uint64_t)v.w[3] | ((uint64_t)v.w[4] << 32));
	st_coh64(q + 0, (uint64_t)v.son0 | ((uint64_t)v.son1 << 32));
}

enum : uint32_t { W_IDLE = 0, W_LOAD = 1, W_SONS = 2, W_FINISH = 3, W_OVER = 4 };

// (Round 3 also kept the tree of buckets of up to 3840 positions in LDS -- k_bt_wave<CAP>: 2.4x less HBM traffic, 4x the
// time on the bench text; tools/experiments/bt_tree_in_lds.patch has that build and its numbers.)
//
// k_bt_group<G>: G lanes of a wavefront per bucket, 64 / G buckets per wavefront (round 4).  G = 64 is the kernel
// above word for word (one wavefront per bucket, for the few buckets of 4096 positions and more).  What the measured
// pipeline depth said -- 7 node visits retire per round whatever the number of lanes -- is that 56 of a wavefront's 64
// lanes wait; and what bounds the lane-per-bucket kernel k_bt is its longest lane (a bucket of 4000 positions is a
// serial chain of ~48 000 dependent node loads).  So the middle of the distribution (on text: 70 % of a block's
// positions sit in buckets of 1024 - 4095) gets EIGHT lanes per bucket: the same pipelined walk, marks and all, with
// everything that was wave-uniform (next position to start, the staged window, run handling) uniform per group of G
// lanes instead -- ballots masked to the group, one staged window of G positions per group -- and the output allocator
// still one prefix sum over the wavefront.  A bucket then moves ~4-5 times faster than on one lane at an eighth of a
// wavefront.
struct StagedWindows { // G positions per group of a wavefront, one column per lane
	uint32_t st_i[64], st_w[5][64], st_c2[64], st_c3[64], st_b[64];
};
template <int G>
__device__ __forceinline__ void bt_group_body(const uint32_t block, StagedWindows &sw, const uint8_t *__restrict__ src, uint32_t n,
					      uint32_t seg_base, uint32_t seg_end, const uint32_t *__restrict__ spos,
					      const uint32_t *__restrict__ seg_len_sorted, const uint32_t *__restrict__ seg_start_sorted,
					      BtNode *node,
					      const uint32_t *__restrict__ prev2, const uint32_t *__restrict__ prev3,
					      uint32_t dict, uint32_t fb, uint32_t cut,
					      uint8_t *__restrict__ counts, uint64_t *__restrict__ tmp_start,
					      uint32_t *__restrict__ pool, unsigned long long *__restrict__ cursor,
					      unsigned long long pool_cap, uint32_t chunk, int *__restrict__ err,
					      unsigned long long *__restrict__ stats, const uint32_t io_min, const uint32_t io_mask)
{
	static_assert(G == 64 || G == 32 || G == 16 || G == 8 || G == 4, "lanes per bucket");
	constexpr int NG = 64 / G;
	const RecColumns rec = rec_columns(cut);
	auto &st_i = sw.st_i;
	auto &st_w = sw.st_w;
	auto &st_c2 = sw.st_c2;
	auto &st_c3 = sw.st_c3;
	auto &st_b = sw.st_b;
	// Orders a round's tree stores before the next round's loads: a workgroup-scope fence (one wave, one CU; an
	// agent-scope fence writes the L2 back on this multi-XCD part: 27 us per round instead of 1.9)
	auto tree_fence = [&]() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); };
	const uint32_t lane = threadIdx.x;
	const uint32_t grp = lane / G, gl = lane % G, gshift = grp * G;
	const uint64_t full_g = G == 64 ? ~(uint64_t)0 : (((uint64_t)1 << (G & 63)) - 1);
	const uint64_t gmask = full_g << gshift;                       // the lanes of this lane's group
	const uint64_t lt_mask = (((uint64_t)1 << lane) - 1) & gmask;  // ... of those, the ones below this lane
	const uint32_t bucket = seg_base + block * NG + grp;
	const bool have = bucket < seg_end;
	const uint32_t k0 = have ? seg_start_sorted[bucket] : 0;
	const uint32_t L = have ? seg_len_sorted[bucket] : 0;
	const uint32_t cyc_size = dict + 1;
	// slot (x, side) = 2 * x + side, x = sorted index: son word `side` of node x (a word index into the node array would
	// leave 32 bits at 2^29 positions; blocks go up to lrzgpu_max_block_bytes())
	auto word_at = [&](uint32_t slot) -> uint32_t * { return reinterpret_cast<uint32_t *>(node + (slot >> 1)) + (slot & 1); };
	auto node_at = [&](uint32_t x) -> BtNode * { return node + x; };
	uint32_t stage_base = 0, stage_end = 0, stage_prev = 0; // group-uniform: the window, and the position before it
	uint32_t pd_c2 = 0, pd_c3 = 0, pd_bytes = 0;            // the walk's h2 / h3 candidates
	// The next window of a group is fetched AHEAD, one level of its chain of dependent loads (place -> prefix and h2 / h3
	// candidates -> the candidates' bytes) per round, issued beside the round's node loads: when the window in use runs
	// out, the next one is in registers and staging it costs no memory round trip (with G = 8 a window lasts ~16 rounds
	// and eight groups share a wavefront: loading it on the spot -- three round trips -- stalled every second round)
	PosData pf{};
	uint32_t pf_base = 0xFFFFFFFFu, pf_stage = 0; // group-uniform: the window pf is for; 0 nothing, 1 place, 2 + prefix, candidates, 3 whole
	auto prefetch_step = [&]() {
		const uint32_t want = stage_end; // (the next window starts where this one ends)
		if (want >= L) {
			pf_stage = 0;
			return;
		}
		const uint32_t q = want + gl;
		if (pf_base != want || pf_stage == 0) {
			pf = PosData{};
			pf_base = want;
			if (q < L)
				pf.i = spos[k0 + q];
			pf_stage = 1;
		} else if (pf_stage == 1) {
			if (q < L) {
				node_prefix(src + pf.i, n - pf.i, pf.w);
				pf.c2 = prev2[pf.i];
				pf.c3 = prev3[pf.i];
			}
			pf_stage = 2;
		} else if (pf_stage == 2) {
			if (q < L)
				pf.bytes = cand_bytes(src, pf.c2, pf.c3);
			pf_stage = 3;
		}
	};
	// stage the G positions of the bucket from `from` on for the groups that `need` it (every lane of such a group
	// holds or loads one; called by the whole wave)
	auto restage = [&](bool need, uint32_t from) {
		tree_fence(); // (the readers of the old window are done)
		if (need) {
			PosData d{};
			uint32_t before; // 1-based position of the bucket's entry before the window (0: none)
			if (from > stage_base && from == stage_end)
				before = st_i[gshift + (from - 1 - stage_base)] + 1; // (the old window's last entry)
			else
				before = from ? spos[k0 + from - 1] + 1 : 0;
			if (pf_stage == 3 && pf_base == from)
				d = pf;
			else {
				const uint32_t q = from + gl;
				if (q < L)
					load_pos(d, src, n, spos, k0 + q, prev2, prev3);
			}
			st_i[lane] = d.i;
#pragma unroll
			for (int k = 0; k < 5; k++)
				st_w[k][lane] = d.w[k];
			st_c2[lane] = d.c2;
			st_c3[lane] = d.c3;
			st_b[lane] = d.bytes;
			stage_prev = before;
			stage_base = from;
			stage_end = from + G < L ? from + G : L;
			pf_stage = 0;
		}
		tree_fence();
	};
	restage(true, 0);
	WaveAlloc wa{0, 0, chunk};
	uint32_t next_j = 0; // group-uniform: the next position of the bucket that has no walk yet

	// per-lane walk state
	uint32_t state = W_IDLE;
	uint32_t wj = 0, i = 0, pos = 0, len_limit = 0, cbs = 0, prev_pos = 0;
	uint32_t mw[5] = {0, 0, 0, 0, 0};
	uint32_t slot0 = 0, slot1 = 0; // ptr0 / ptr1 of the reference as word indices
	uint32_t len0 = 0, len1 = 0, max_len = 3, cv = 0, cur_ref = 0, nrec = 0;
	// the node the walk stands on, kept while it waits for a son
	uint32_t n_len = 0;  // agreement with it
	uint32_t n_side = 0; // 0 / 1: the son it will descend into; 2: full-length agreement, both sons are taken over
	bool run_hit = false;
	uint32_t run_s0 = 0, run_s1 = 0;
	uint32_t st_rounds = 0, st_steps = 0, st_stalls = 0; // (st_rounds wave-uniform, the others per lane)

	for (;;) {
		st_rounds++;
		// Starting walks and writing finished ones out are the long, thinly populated parts of a round (a handful of the
		// 64 lanes each): they run when enough lanes wait for them, or every few rounds, not in every round.  A finished
		// walk has resolved its slots already (nothing waits for its lists), an idle lane only lowers the parallelism.
		bool do_io = (st_rounds & io_mask) == 1 || (uint32_t)__popcll(__ballot(state == W_IDLE || state == W_FINISH)) >= io_min;
		// ---- start walks on free lanes, in bucket order --------------------------------------------------------
		// (what a walk needs of its position comes from its group's staged window: G positions loaded at once)
		if (do_io) {
			const bool idle = state == W_IDLE;
			const uint64_t m = __ballot(idle) & gmask;
			if (m) { // (group-uniform)
				const uint32_t rank = (uint32_t)__popcll(m & lt_mask);
				const uint32_t j = next_j + rank;
				const uint32_t room = stage_end - next_j; // positions of the staged window not handed out yet
				if (idle && next_j >= L)
					state = W_OVER;
				else if (idle && rank < room && j < L) {
					const uint32_t q = gshift + (j - stage_base);
					wj = j;
					const uint32_t self = k0 + j;
					i = st_i[q];
#pragma unroll
					for (int k = 0; k < 5; k++)
						mw[k] = st_w[k][q];
					pd_c2 = st_c2[q];
					pd_c3 = st_c3[q];
					pd_bytes = st_b[q];
					pos = i + 1;
					prev_pos = j != stage_base ? st_i[q - 1] + 1 : stage_prev;
					const uint32_t avail = n - i;
					len_limit = avail < fb ? avail : fb;
					cbs = pos < cyc_size ? pos : cyc_size;
					nrec = 0;
					run_hit = false;
					BtNode me;
					me.pos = pos;
#pragma unroll
					for (int k = 0; k < 5; k++)
						me.w[k] = mw[k];
					if (pos - prev_pos >= cbs) { // (prev_pos == 0: the bucket's first position)
						me.son0 = me.son1 = 0;
						state = W_FINISH;
					} else {
						me.son0 = me.son1 = kPending;
						slot0 = 2 * self + 1; // ptr0 = &son1, ptr1 = &son0
						slot1 = 2 * self;
						len0 = len1 = 0;
						max_len = 3;
						cv = cut;
						cur_ref = self; // sorted index + 1 of the predecessor
						state = W_LOAD;
					}
					store_node_coh(node_at(self), me);
				}
				uint32_t started = (uint32_t)__popcll(m);
				if (started > room)
					started = room;
				next_j += started;
				if (next_j > L)
					next_j = L;
			}
			// a group's window is used up: stage the next G positions of its bucket
			const bool need = next_j == stage_end && next_j < L;
			if (__any(need))
				restage(need, next_j);
		}
		if (__ballot(state != W_OVER) == 0)
			break;
		st_steps += (state == W_LOAD) ? 1u : 0u;
		st_stalls += (state == W_SONS) ? 1u : 0u;
		// the stores of this round (new nodes, resolved and marked slots) complete before the loads of the next
		tree_fence();
		if (G < 64 && __any(stage_end < L && !(pf_stage == 3 && pf_base == stage_end)))
			prefetch_step();

		// ---- one step of every walk -----------------------------------------------------------------------------
		if (state == W_LOAD || state == W_SONS) {
			const uint32_t x = cur_ref - 1;
			const uint64_t *q = reinterpret_cast<const uint64_t *>(node_at(x));
			const uint64_t sons = ld_coh64(q);
			uint32_t s0 = (uint32_t)sons, s1 = (uint32_t)(sons >> 32);
			bool go = true;
			if (state == W_LOAD) {
				const uint64_t a = ld_coh64(q + 1), b = ld_coh64(q + 2), c = ld_coh64(q + 3);
				const uint32_t npos = (uint32_t)a;
				const uint32_t nw[5] = {(uint32_t)(a >> 32), (uint32_t)b, (uint32_t)(b >> 32), (uint32_t)c, (uint32_t)(c >> 32)};
				const uint32_t delta = pos - npos;
				if (delta >= cbs) {
					st_coh(word_at(slot0), 0);
					st_coh(word_at(slot1), 0);
					state = W_FINISH;
					go = false;
				} else {
					const uint8_t *cur = src + i, *pb = cur - delta;
					const uint32_t len = agree_len(nw, mw, cur, pb, len0 < len1 ? len0 : len1, len_limit);
					n_len = len;
					n_side = 0;
					if (max_len < len) {
						max_len = len;
						rec.put(nrec >> 1, lane, len, delta - 1);
						nrec += 2;
						if (len == len_limit) {
							n_side = 2;
							run_hit = delta == 1 && nrec == 2 && len_limit == fb && pos == prev_pos + 1;
						}
					}
					if (n_side != 2) {
						uint32_t b_node, b_cur;
						if (len < kNodeBytes) {
							b_node = prefix_byte(nw, len);
							b_cur = prefix_byte(mw, len);
						} else {
							b_node = pb[len];
							b_cur = cur[len];
						}
						n_side = b_node < b_cur ? 1 : 0;
					}
					state = W_SONS;
				}
			}
			if (go) {
				if (n_side == 2) {
					if (s0 != kPending && s1 != kPending) {
						st_coh(word_at(slot1), s0);
						st_coh(word_at(slot0), s1);
						run_s0 = s0;
						run_s1 = s1;
						state = W_FINISH;
					} // else: waits; the verdict stands
				} else {
					const uint32_t next = n_side ? s1 : s0;
					if (next != kPending) {
						const uint32_t taken = 2 * x + n_side; // &x.son1 (smaller) / &x.son0
						if (n_side) {
							st_coh(word_at(slot1), cur_ref);
							slot1 = taken;
							len1 = n_len;
						} else {
							st_coh(word_at(slot0), cur_ref);
							slot0 = taken;
							len0 = n_len;
						}
						if (next >= cur_ref) { // corrupt tree (cannot happen): stop like the reference
							*err = 2;
							st_coh(word_at(slot0), 0);
							st_coh(word_at(slot1), 0);
							state = W_FINISH;
						} else if (--cv == 0 || next == 0) {
							st_coh(word_at(slot0), 0);
							st_coh(word_at(slot1), 0);
							state = W_FINISH;
						} else {
							st_coh(word_at(taken), kPending); // ours until this walk writes its next subtree root there
							cur_ref = next;
							state = W_LOAD;
						}
					}
				}
			}
		}

		// ---- a run of one byte value at the head of a group's pipeline: G positions per round ----------------------------
		// (the walk that found it is the oldest in flight: everything younger stands at its first node, waiting for a
		//  son of its predecessor, and has written nothing but its own node -- those walks are simply started again)
		uint32_t run_from = L; // group-uniform after the ballots below
		{
			const bool hit = state == W_FINISH && run_hit && len_limit == fb;
			const uint64_t hm = __ballot(hit) & gmask;
			const int src_lane = hm ? __ffsll((long long)hm) - 1 : (int)lane;
			const uint32_t hj = __shfl(wj, src_lane);
			// older walks still in flight?
			const uint64_t older = __ballot(hm != 0 && (state == W_LOAD || state == W_SONS || state == W_FINISH) && wj < hj) & gmask;
			if (hm && !older && __popcll(hm) == 1)
				run_from = hj;
		}
		const bool in_run = run_from != L;
		if (__any(in_run))
			do_io = true; // (the run path below starts every lane of its group again: what is finished goes out first)
		// a full-length first step that is not taken as a run NOW is never one: a round later the walks behind it may
		// have moved on (its node's sons are resolved), and a finished walk may wait several rounds for its write-out
		if (run_hit && state == W_FINISH && !(in_run && wj == run_from))
			run_hit = false;

		// ---- finished walks: h2 / h3 candidates, lists out -----------------------------------------------------------------
		if (do_io) {
			const bool fin = state == W_FINISH && (run_from == L || wj <= run_from);
			uint32_t mix[4];
			uint32_t nmix = 0;
			if (fin) {
				PosData d;
				d.w[0] = mw[0];
				d.c2 = pd_c2;
				d.c3 = pd_c3;
				d.bytes = pd_bytes;
				nmix = mix_from(d, pos, dict, nrec != 0, nrec ? rec.entry(1, lane) : 0, mix);
			}
			const uint32_t cnt = fin ? nmix + nrec : 0;
			const unsigned long long st = wave_take(wa, cnt, cursor);
			if (fin) {
				counts[i] = (uint8_t)cnt;
				if (cnt) {
					tmp_start[i] = st;
					if (st + cnt > pool_cap)
						*err = 1;
					else {
						uint32_t *o = pool + st;
						put_mix(o, mix, nmix);
						for (uint32_t k = 0; k < nrec; k++)
							o[nmix + k] = rec.entry(k, lane);
					}
				}
				state = W_IDLE;
			}
		}

		if (__any(in_run)) {
			// the lane of walk run_from holds the run's byte and sons
			const uint64_t om = __ballot(in_run && wj == run_from && run_hit) & gmask;
			const int ol = om ? __ffsll((long long)om) - 1 : (int)lane;
			const uint32_t ri = __shfl(i, ol), rs0 = __shfl(run_s0, ol), rs1 = __shfl(run_s1, ol);
			const uint32_t b = in_run ? src[ri + fb - 1] : 0;
			uint32_t t = 0;
			bool going = in_run;
			while (__any(going)) {
				const uint32_t idx = run_from + 1 + t + gl; // position of the bucket this lane looks at
				const uint32_t iq = ri + t + 1 + gl;
				bool ok = going && idx < L && (unsigned long long)iq + fb <= n;
				if (ok)
					ok = spos[k0 + idx] == iq && src[iq + fb - 1] == b;
				const uint64_t okm = (__ballot(ok) & gmask) >> gshift;
				const uint32_t m = okm == full_g ? (uint32_t)G : (uint32_t)(__ffsll((long long)~okm) - 1); // leading run of ok lanes
				const bool mine = going && gl < m;
				const unsigned long long st = wave_take(wa, mine ? 2u : 0u, cursor);
				if (mine) {
					BtNode rn;
					rn.son0 = rs0;
					rn.son1 = rs1;
					rn.pos = iq + 1;
#pragma unroll
					for (int k = 0; k < 5; k++)
						rn.w[k] = b * 0x01010101u; // fb >= 20 bytes of the run lie ahead of every one of them
					store_node_coh(node_at(k0 + idx), rn);
					counts[iq] = 2;
					tmp_start[iq] = st;
					if (st + 2 > pool_cap)
						*err = 1;
					else {
						pool[st] = fb;
						pool[st + 1] = 0;
					}
				}
				if (going) {
					t += m;
					if (m < (uint32_t)G)
						going = false;
				}
			}
			// everything younger of that group starts again behind the run
			if (in_run) {
				next_j = run_from + 1 + t;
				if (state != W_OVER || next_j < L)
					state = W_IDLE;
				run_hit = false;
			}
			const bool need = in_run && next_j < L;
			if (__any(need))
				restage(need, next_j);
		}
	}
	// how the pipeline did (tools/bt_case.py): rounds of this wave, node visits, rounds a walk spent waiting for a son
	uint32_t st_len = have && gl == 0 ? L : 0;
	for (int o = 32; o; o >>= 1) {
		st_steps += __shfl_down(st_steps, o);
		st_stalls += __shfl_down(st_stalls, o);
		st_len += __shfl_down(st_len, o);
	}
	if (lane == 0) {
		atomicAdd(stats + 0, (unsigned long long)st_rounds);
		atomicAdd(stats + 1, (unsigned long long)st_steps);
		atomicAdd(stats + 2, (unsigned long long)st_stalls);
		atomicAdd(stats + 3, (unsigned long long)st_len);
	}
}

// One launch for the three ways a bucket is walked, longest buckets first (workgroups are dispatched in index order, so
// the few wavefront-per-bucket chains -- the launch's critical path -- start at once and the rest of the chip works
// through the eight-lane groups and the lane-per-bucket waves beside them; as three launches on one stream they ran one
// after the other: 17.8 + 41.6 + 8.1 ms on a 64 MiB block of the bench text).
__global__ void __launch_bounds__(64) k_bt_walk(const uint8_t *__restrict__ src, uint32_t n, const uint32_t *__restrict__ spos,
						const uint32_t *__restrict__ seg_len_sorted, const uint32_t *__restrict__ seg_start_sorted,
						uint32_t nseg, uint32_t nlong, uint32_t nmid_end, BtNode *node,
						const uint32_t *__restrict__ prev2, const uint32_t *__restrict__ prev3,
						uint32_t dict, uint32_t fb, uint32_t cut,
						uint8_t *__restrict__ counts, uint64_t *__restrict__ tmp_start,
						uint32_t *__restrict__ pool, unsigned long long *__restrict__ cursor,
						unsigned long long pool_cap, uint32_t chunk, int *__restrict__ err,
						unsigned long long *__restrict__ stats, uint32_t io_min, uint32_t io_mask)
{
	__shared__ StagedWindows sw;
	const uint32_t ngrp_waves = (nmid_end - nlong + 7) / 8;
	if (blockIdx.x < nlong)
		bt_group_body<64>(blockIdx.x, sw, src, n, 0u, nlong, spos, seg_len_sorted, seg_start_sorted, node, prev2, prev3, dict, fb, cut, counts,
				  tmp_start, pool, cursor, pool_cap, chunk, err, stats, 1u, 0u);
	else if (blockIdx.x < nlong + ngrp_waves)
		bt_group_body<8>(blockIdx.x - nlong, sw, src, n, nlong, nmid_end, spos, seg_len_sorted, seg_start_sorted, node, prev2, prev3, dict, fb, cut,
				 counts, tmp_start, pool, cursor, pool_cap, chunk, err, stats, io_min, io_mask);
	else
		bt_lane_body(blockIdx.x - nlong - ngrp_waves, src, n, spos, seg_len_sorted, seg_start_sorted, nseg, nmid_end, node, prev2, prev3, dict, fb,
			     cut, counts, tmp_start, pool, cursor, pool_cap, chunk, err);
}

// ---------------------------------------------------------------------------------------------
// HC5: the hash-chain finder of LZMA levels 1-4 (algo 0: btMode 0, 5 hash bytes, single-threaded in
// the reference: Hc5_MatchFinder_GetMatches / _Skip, LzFind.c:1431-1502, 1619-1649, and
// Hc_GetMatchesSpec, LzFind.c:880-958).  GetMatches and Skip update the three hash tables and the
// chain identically at every position that still has 5 bytes, so the chain link of a position is
// simply "the previous position with the same 5-byte hash" and its h2/h3 candidates "the previous
// position with the same 10/16-bit hash": all of them come from stable sorts, and, unlike the binary
// tree, nothing a position does changes what a later one sees.  One thread per position.
// ---------------------------------------------------------------------------------------------
// Records live in LDS columns (RecColumns: 2 hash pairs + at most `cut` chain pairs per lane) -- as a register array indexed
// by a variable they were 288 bytes of scratch memory per lane -- and a wavefront takes its output space with one atomic
// (prefix sum over the lanes' counts) instead of one per position.
__global__ void __launch_bounds__(64) k_hc5(const uint8_t *__restrict__ src, uint32_t n, const uint32_t *__restrict__ prev2,
					    const uint32_t *__restrict__ prev3, const uint32_t *__restrict__ prev5, uint32_t dict,
					    uint32_t fb, uint32_t cut, uint8_t *__restrict__ counts, uint64_t *__restrict__ tmp_start,
					    uint32_t *__restrict__ pool, unsigned long long *__restrict__ cursor,
					    unsigned long long pool_cap, int *__restrict__ err)
{
	const RecColumns rec = rec_columns(cut + 2);
	const uint32_t lane = threadIdx.x;
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	const uint32_t avail = i < n ? n - i : 0;
	const uint32_t len_limit = avail < fb ? avail : fb;
	const bool act = len_limit >= 5; // GET_MATCHES_HEADER(5): a position with fewer bytes is skipped, no table is touched
	uint32_t np = 0;                 // pairs recorded
	if (act) {
		const uint32_t pos = i + 1;
		const uint8_t *cur = src + i;
		const uint32_t cbs = dict + 1;
		const uint32_t mmm = pos < cbs ? pos : cbs; // SET_mmm
		uint32_t d2 = pos - prev2[i], d3 = pos - prev3[i];
		uint32_t max_len = 4;
		bool chain = true;
		{
			bool have = false; // the last pair's length is still to be written
			if (d2 < mmm && cur[-(int64_t)d2] == cur[0]) {
				rec.put(np++, lane, 2, d2 - 1);
				if (cur[2 - (int64_t)d2] == cur[2]) {
					have = true;
				} else if (d3 < mmm && cur[-(int64_t)d3] == cur[0]) {
					rec.put(np++, lane, 0, d3 - 1);
					d2 = d3;
					have = true;
				}
			} else if (d3 < mmm && cur[-(int64_t)d3] == cur[0]) {
				rec.put(np++, lane, 0, d3 - 1);
				d2 = d3;
				have = true;
			}
			if (have) {
				uint32_t l3 = 3;
				if (cur[3 - (int64_t)d2] == cur[3]) {
					uint32_t l = max_len; // UPDATE_maxLen: from byte 4 on
					while (l != len_limit && cur[l - (int64_t)d2] == cur[l])
						l++;
					max_len = l;
					l3 = max_len;
					if (max_len == len_limit)
						chain = false;
				}
				rec.len[(np - 1) * 64 + lane] = (uint16_t)l3;
			}
		}
		if (chain) {
			uint32_t cm = prev5[i], cv = cut;
			do {
				if (cm == 0)
					break;
				const uint32_t delta = pos - cm;
				if (delta >= cbs)
					break;
				const uint32_t next = prev5[cm - 1]; // the chain link that position stored
				const uint8_t *pb = cur - delta;
				if (cur[max_len] == pb[max_len]) {
					uint32_t len = 0;
					while (len != len_limit && cur[len] == pb[len])
						len++;
					if (len == len_limit) {
						rec.put(np++, lane, len_limit, delta - 1);
						break;
					}
					if (max_len < len) {
						max_len = len;
						rec.put(np++, lane, len, delta - 1);
					}
				}
				cm = next;
			} while (--cv);
		}
	}
	const uint32_t nrec = 2 * np;
	// output space: one atomic per wavefront
	uint32_t incl = nrec;
#pragma unroll
	for (int o = 1; o < 64; o <<= 1) {
		const uint32_t v = __shfl_up(incl, o);
		if ((int)lane >= o)
			incl += v;
	}
	const uint32_t total = __shfl(incl, 63);
	unsigned long long base = 0;
	if (total) {
		if (lane == 0)
			base = atomicAdd(cursor, (unsigned long long)total);
		base = __shfl(base, 0);
	}
	if (act) {
		counts[i] = (uint8_t)nrec;
		if (nrec) {
			const unsigned long long st = base + (incl - nrec);
			tmp_start[i] = st;
			if (st + nrec > pool_cap)
				*err = 1;
			else
				for (uint32_t k = 0; k < nrec; k++)
					pool[st + k] = rec.entry(k, lane);
		}
	}
}

struct CountToU64 {
	__host__ __device__ unsigned long long operator()(const uint8_t &c) const { return (unsigned long long)c; }
};

// Output formats of the position-ordered lists (`mode`):
//   0  plain: (len, dist-1) u32 couples, as MatchFinderMt_GetMatches returns them
//   1  flagged: the same with bit 31 of `len` = the tail flag
//   2  packed + flagged: one u32 per pair = flag << 31 | (len - 2) << 25 | dist-1 (dictionaries up to 32 MiB,
//      len <= 65): halves the PCIe volume of the lists; the host parser reads them in place
// The tail flag answers the one question the optimal parser asks about the block's bytes at EVERY candidate
// distance of EVERY position -- "after this match and one more (literal) byte, do the next two bytes continue at
// the same distance?" (the match + literal + repeat-0 candidate, reference LzmaEnc.c:1900-1960).  On the host
// that is a random access up to a dictionary behind per pair, a cache miss each, and the answer is "no" 99.9%
// of the time; here it is two more loads of a thread that holds the pair anyway.
__device__ inline uint32_t tail_flag(const uint8_t *__restrict__ src, uint32_t n, uint32_t pos, uint32_t len, uint32_t dist1)
{
	const uint64_t a = (uint64_t)pos + len + 1; // first byte after the literal
	if (a + 2 > n)
		return 0;
	const uint8_t *p = src + a, *q = p - dist1 - 1;
	return (p[0] == q[0] && p[1] == q[1]) ? 1u : 0u;
}

__global__ void __launch_bounds__(256) k_gather(const uint8_t *__restrict__ counts, const uint64_t *__restrict__ tmp_start,
						const unsigned long long *__restrict__ offsets, const uint32_t *__restrict__ pool,
						uint32_t *__restrict__ out, uint32_t n, unsigned long long pool_cap, int mode,
						const uint8_t *__restrict__ src)
{
	uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	uint32_t stride = gridDim.x * blockDim.x;
	for (; i < n; i += stride) {
		uint32_t c = counts[i];
		if (!c || tmp_start[i] + c > pool_cap || offsets[i] + c > pool_cap)
			continue;
		const uint32_t *s = pool + tmp_start[i];
		if (mode == 2) {
			uint32_t *d = out + (offsets[i] >> 1);
			for (uint32_t k = 0; k < c; k += 2)
				d[k >> 1] = (tail_flag(src, n, i, s[k], s[k + 1]) << 31) | ((s[k] - 2) << 25) | s[k + 1];
		} else if (mode == 1) {
			uint32_t *d = out + offsets[i];
			for (uint32_t k = 0; k < c; k += 2) {
				d[k] = s[k] | (tail_flag(src, n, i, s[k], s[k + 1]) << 31);
				d[k + 1] = s[k + 1];
			}
		} else {
			uint32_t *d = out + offsets[i];
			for (uint32_t k = 0; k < c; k++)
				d[k] = s[k];
		}
	}
}

__global__ void k_total(const uint8_t *counts, const unsigned long long *offsets, uint32_t n, unsigned long long *total)
{
	*total = n ? offsets[n - 1] + counts[n - 1] : 0;
}

// ---------------------------------------------------------------------------------------------

int mf_workspace_create(MfWorkspace **out, size_t max_n, double pool_per_pos)
{
	MfWorkspace *w = (MfWorkspace *)calloc(1, sizeof(MfWorkspace));
	if (!w)
		return -1;
	w->max_n = max_n;
	w->pool_cap = (unsigned long long)((double)max_n * pool_per_pos) + 4096;
	size_t n = max_n + 8;
	HIPCHK(hipMalloc(&w->key_a, n * 4));
	HIPCHK(hipMalloc(&w->key_b, n * 4));
	HIPCHK(hipMalloc(&w->val_a, n * 4));
	HIPCHK(hipMalloc(&w->val_b, n * 4));
	HIPCHK(hipMalloc(&w->spos, n * 4));
	HIPCHK(hipMalloc(&w->prev2, n * 4));
	HIPCHK(hipMalloc(&w->prev3, n * 4));
	HIPCHK(hipMalloc(&w->seg_start, n * 4));
	HIPCHK(hipMalloc(&w->seg_len, n * 4));
	HIPCHK(hipMalloc(&w->seg_start_s, n * 4));
	HIPCHK(hipMalloc(&w->seg_len_s, n * 4));
	HIPCHK(hipMalloc(&w->flags, n));
	HIPCHK(hipMalloc(&w->son, (n + 2) * 32)); // BtNode per sorted index (k_bt); u32 per position for k_hc5
	HIPCHK(hipMalloc(&w->counts, n));
	HIPCHK(hipMalloc(&w->tmp_start, n * 8));
	HIPCHK(hipMalloc(&w->offsets, n * 8));
	HIPCHK(hipMalloc(&w->pool_tmp, w->pool_cap * 4));
	HIPCHK(hipMalloc(&w->pool_out, w->pool_cap * 4));
	HIPCHK(hipMalloc(&w->scalars, 128));
	// temp storage: the largest request among the rocPRIM calls used below
	size_t need = 0, t = 0;
	(void)rocprim::radix_sort_pairs(nullptr, t, w->key_a, w->key_b, w->val_a, w->val_b, (size_t)max_n, 0, 32);
	need = t;
	(void)rocprim::radix_sort_pairs_desc(nullptr, t, w->key_a, w->key_b, w->val_a, w->val_b, (size_t)max_n, 0, 32);
	if (t > need) need = t;
	(void)rocprim::select(nullptr, t, rocprim::counting_iterator<uint32_t>(0), w->flags, w->seg_start,
				      (uint32_t *)w->scalars, (size_t)max_n);
	if (t > need) need = t;
	rocprim::transform_iterator<const uint8_t *, CountToU64, unsigned long long> it(w->counts, CountToU64());
	(void)rocprim::exclusive_scan(nullptr, t, it, (unsigned long long *)w->offsets, 0ull, (size_t)max_n, rocprim::plus<unsigned long long>());
	if (t > need) need = t;
	w->prim_bytes = need + 256;
	HIPCHK(hipMalloc(&w->prim_tmp, w->prim_bytes));
	*out = w;
	return 0;
}

void mf_workspace_destroy(MfWorkspace *w)
{
	if (!w)
		return;
	void *ptrs[] = {w->key_a, w->key_b, w->val_a, w->val_b, w->spos, w->prev2, w->prev3, w->seg_start, w->seg_len,
			w->seg_start_s, w->seg_len_s, w->flags, w->son, w->counts, w->tmp_start, w->offsets, w->pool_tmp,
			w->pool_out, w->scalars, w->prim_tmp};
	for (void *p : ptrs)
		if (p)
			(void)hipFree(p);
	free(w);
}

static inline int grid_for(size_t n, int block) // ~8 blocks per CU, grid-stride beyond
{
	size_t g = (n + block - 1) / block;
	if (g > 256 * 8)
		g = 256 * 8;
	if (g == 0)
		g = 1;
	return (int)g;
}

// Runs the finder on d_src[0..n) (device). Results stay on the device in w->counts / w->pool_out;
// *total_entries receives the number of u32 entries.
int mf_run_device(MfWorkspace *w, const uint8_t *d_src, size_t n, uint32_t dict, uint32_t fb, uint32_t cut,
		  hipStream_t s, unsigned long long *total_entries, int mode, bool hc5, size_t block_n)
{
	const size_t mask_n = block_n > n ? block_n : n; // the size the reference's finder would be created for
	if (mode < 0 || mode > 2 || (mode == 2 && (dict > (1u << 25) || fb > 65)))
		return -3;
	if (n > w->max_n || n >= 0x7FFFFFF0ull) // (son slots are 32-bit `2 * node + side` words: lrzgpu_max_block_bytes() has the same cap)
		return -2;
	if (2 * (cut + 2) > (uint32_t)kMaxRec || 2 * (cut + 2) > 255)
		return -3;
	unsigned long long *d_cursor = (unsigned long long *)w->scalars;      // [0]
	unsigned long long *d_total = (unsigned long long *)w->scalars + 1;   // [1]
	uint32_t *d_nseg = (uint32_t *)((unsigned long long *)w->scalars + 2); // [2]
	int *d_err = (int *)((unsigned long long *)w->scalars + 3);           // [3]
	EventTimer t_all(s);
	EventTimer *t_bt = nullptr;
	HIPCHK(hipMemsetAsync(w->scalars, 0, 128, s));
	uint32_t *d_nlong = (uint32_t *)((unsigned long long *)w->scalars + 8); // [8]: buckets for the wave-per-bucket kernel
	*total_entries = 0;
	if (n == 0)
		return 0;
	HIPCHK(hipMemsetAsync(w->counts, 0, n, s));
	if (hc5) {
		if (cut > 32)
			return -3;
		if (n >= 5) {
			const uint32_t n5 = (uint32_t)(n - 4);
			const uint32_t mask = lzma_hash_mask5(dict, mask_n);
			const int bits = 32 - __builtin_clz(mask);
			size_t tb;
			const int g = grid_for(n5, 256);
			uint32_t *prev5 = w->son; // the tree array is free in this mode
			hipLaunchKernelGGL(k_keys<2>, dim3(g), dim3(256), 0, s, d_src, n5, mask, 0, w->key_a, w->val_a);
			tb = w->prim_bytes;
			HIPCHK(rocprim::radix_sort_pairs(w->prim_tmp, tb, w->key_a, w->key_b, w->val_a, w->val_b, (size_t)n5, 0, 10, s));
			hipLaunchKernelGGL(k_link_prev, dim3(g), dim3(256), 0, s, w->key_b, w->val_b, n5, w->prev2);
			hipLaunchKernelGGL(k_keys<3>, dim3(g), dim3(256), 0, s, d_src, n5, mask, 0, w->key_a, w->val_a);
			tb = w->prim_bytes;
			HIPCHK(rocprim::radix_sort_pairs(w->prim_tmp, tb, w->key_a, w->key_b, w->val_a, w->val_b, (size_t)n5, 0, 16, s));
			hipLaunchKernelGGL(k_link_prev, dim3(g), dim3(256), 0, s, w->key_b, w->val_b, n5, w->prev3);
			hipLaunchKernelGGL(k_keys<5>, dim3(g), dim3(256), 0, s, d_src, n5, mask, 0, w->key_a, w->val_a);
			tb = w->prim_bytes;
			HIPCHK(rocprim::radix_sort_pairs(w->prim_tmp, tb, w->key_a, w->key_b, w->val_a, w->val_b, (size_t)n5, 0, bits, s));
			hipLaunchKernelGGL(k_link_prev, dim3(g), dim3(256), 0, s, w->key_b, w->val_b, n5, prev5);
			t_bt = new EventTimer(s);
			hipLaunchKernelGGL(k_hc5, dim3((unsigned)((n + 63) / 64)), dim3(64), rec_lds_bytes(cut + 2), s, d_src, (uint32_t)n, w->prev2, w->prev3, prev5, dict, fb,
					   cut, w->counts, w->tmp_start, w->pool_tmp, d_cursor, w->pool_cap, d_err);
			t_bt->stop();
		}
	} else if (n >= 4) {
		const uint32_t n4 = (uint32_t)(n - 3);
		const uint32_t mask = lzma_hash_mask(dict, mask_n);
		const int big = mask >= 0xFFFFFF;
		int bits = 32 - __builtin_clz(mask);
		size_t tb;
		const int g = grid_for(n4, 256);

		// h2 table: previous position with the same 10-bit hash
		hipLaunchKernelGGL(k_keys<2>, dim3(g), dim3(256), 0, s, d_src, n4, mask, big, w->key_a, w->val_a);
		tb = w->prim_bytes;
		HIPCHK(rocprim::radix_sort_pairs(w->prim_tmp, tb, w->key_a, w->key_b, w->val_a, w->val_b, (size_t)n4, 0, 10, s));
		hipLaunchKernelGGL(k_link_prev, dim3(g), dim3(256), 0, s, w->key_b, w->val_b, n4, w->prev2);
		// h3 table
		hipLaunchKernelGGL(k_keys<3>, dim3(g), dim3(256), 0, s, d_src, n4, mask, big, w->key_a, w->val_a);
		tb = w->prim_bytes;
		HIPCHK(rocprim::radix_sort_pairs(w->prim_tmp, tb, w->key_a, w->key_b, w->val_a, w->val_b, (size_t)n4, 0, 16, s));
		hipLaunchKernelGGL(k_link_prev, dim3(g), dim3(256), 0, s, w->key_b, w->val_b, n4, w->prev3);
		// main hash: buckets
		hipLaunchKernelGGL(k_keys<4>, dim3(g), dim3(256), 0, s, d_src, n4, mask, big, w->key_a, w->val_a);
		tb = w->prim_bytes;
		HIPCHK(rocprim::radix_sort_pairs(w->prim_tmp, tb, w->key_a, w->key_b, w->val_a, w->spos, (size_t)n4, 0, bits, s));
		hipLaunchKernelGGL(k_flag_heads, dim3(g), dim3(256), 0, s, w->key_b, n4, w->flags);
		tb = w->prim_bytes;
		HIPCHK(rocprim::select(w->prim_tmp, tb, rocprim::counting_iterator<uint32_t>(0), w->flags,
						     w->seg_start, d_nseg, (size_t)n4, s));
		// buckets of at least long_min positions get a wavefront each, those of at least mid_min eight lanes (k_bt_group),
		// the rest one lane (k_bt); read per call: tests force 1 (every bucket through the pipelined kernels) and a huge
		// value (none) inside one process
		// LRZGPU_BT_MIN=<wave>[,<group>]
		uint32_t long_min = 4096, mid_min = 512;
		if (const char *e = getenv("LRZGPU_BT_MIN")) {
			auto clip = [](long v) { return (uint32_t)(v < 1 ? 1 : (v > 0x7FFFFFFF ? 0x7FFFFFFF : v)); };
			char *rest = nullptr;
			long_min = clip(strtol(e, &rest, 10));
			if (rest && *rest == ',')
				mid_min = clip(strtol(rest + 1, nullptr, 10));
		}
		if (mid_min > long_min)
			mid_min = long_min;
		hipLaunchKernelGGL(k_seg_len, dim3(g), dim3(256), 0, s, w->seg_start, d_nseg, n4, w->seg_len, long_min, mid_min, d_nlong);
		uint32_t sc[14] = {0};
		HIPCHK(d2h_pageable(sc, d_nseg, 56, s)); // nseg at [0], the long buckets at [12], long + middle at [13]  (sleeps while the sorts run)
		const uint32_t nseg = sc[0];
		const uint32_t nlong = sc[12] > nseg ? nseg : sc[12];
		const uint32_t nmid_end = sc[13] > nseg ? nseg : (sc[13] < nlong ? nlong : sc[13]);
		const uint32_t ngrp_waves = (nmid_end - nlong + 7) / 8;
		// longest buckets first
		tb = w->prim_bytes;
		HIPCHK(rocprim::radix_sort_pairs_desc(w->prim_tmp, tb, w->seg_len, w->seg_len_s, w->seg_start,
								    w->seg_start_s, (size_t)nseg, 0, 32, s));
		const uint32_t chunk = pool_chunk(w->pool_cap, (unsigned long long)nlong + ngrp_waves + (nseg - nmid_end + 63) / 64);
		t_bt = new EventTimer(s);
		// the eight-lane kernel starts walks / writes finished ones out when 24 lanes of the wavefront wait for it, at the
		// latest every 8th round (measured on a 64 MiB block of the bench text: every round 51.2 ms, 12 / 4: 47.8,
		// 24 / 8: 46.5, 32 / 16: 45.9 -- rounds get cheaper faster than they get more)
		const uint32_t io_min = 24, io_mask = 8 - 1;
		const uint32_t nblocks = nlong + ngrp_waves + (nseg - nmid_end + 63) / 64;
		// LDS per workgroup: the record columns, but never so little that a walk workgroup fits beside a resolver on its CU
		// (k_resolve_mw<4> holds 136 of the 160 KB; a finder wave sharing its SIMDs is what slowed the scans down when the
		// wave-per-bucket launches ran beside k_bt in round 3)
		size_t walk_lds = rec_lds_bytes(cut);
		if (walk_lds + sizeof(StagedWindows) < ((size_t)26 << 10))
			walk_lds = ((size_t)26 << 10) - sizeof(StagedWindows);
		if (nblocks)
			hipLaunchKernelGGL(k_bt_walk, dim3(nblocks), dim3(64), walk_lds, s, d_src, (uint32_t)n, w->spos, w->seg_len_s, w->seg_start_s,
					   nseg, nlong, nmid_end, (BtNode *)w->son, w->prev2, w->prev3, dict, fb, cut, w->counts, w->tmp_start, w->pool_tmp,
					   d_cursor, w->pool_cap, chunk, d_err, (unsigned long long *)w->scalars + 4, io_min, io_mask);
		t_bt->stop();
	}
	{
		size_t tb = w->prim_bytes;
		rocprim::transform_iterator<const uint8_t *, CountToU64, unsigned long long> it(w->counts, CountToU64());
		HIPCHK(rocprim::exclusive_scan(w->prim_tmp, tb, it, (unsigned long long *)w->offsets, 0ull, (size_t)n, rocprim::plus<unsigned long long>(), s));
		hipLaunchKernelGGL(k_gather, dim3(grid_for(n, 256)), dim3(256), 0, s, w->counts, w->tmp_start,
				   (const unsigned long long *)w->offsets, w->pool_tmp, w->pool_out, (uint32_t)n, w->pool_cap, mode, d_src);
		hipLaunchKernelGGL(k_total, dim3(1), dim3(1), 0, s, w->counts, (const unsigned long long *)w->offsets, (uint32_t)n, d_total);
	}
	t_all.stop();
	unsigned long long host_sc[8];
	HIPCHK(d2h_pageable(host_sc, w->scalars, 64, s)); // (sleeps while the walks and the gather run)
	{
		ProfileStore &ps = ProfileStore::get();
		std::lock_guard<std::mutex> lk(ps.mu);
		ps.p.mf_total_ms += t_all.ms_noted(ps, PK_MF_TOTAL);
		if (t_bt)
			ps.p.mf_bt_ms += t_bt->ms_noted(ps, PK_MF_BT);
		ps.p.mf_launches++;
		ps.p.mf_positions += (int64_t)n;
		ps.p.mf_entries += (int64_t)host_sc[1];
		for (int k = 0; k < 4; k++)
			ps.p.mf_wave_dbg[k] += (int64_t)host_sc[4 + k];
	}
	delete t_bt;
	int err = (int)(host_sc[3] & 0xFFFFFFFFu);
	if (err == 1)
		return -4; // pool too small
	if (err)
		return -5;
	*total_entries = host_sc[1];
	return 0;
}

} // namespace lrzgpu
