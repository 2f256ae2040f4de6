// lzma_parser.cpp -- host half of the LZMA back end: price-driven parse of a block into literals / matches /
// repeats from the GPU's per-position match lists, and the range coding of that parse.
//
// The bytes must equal what the reference encoder writes (src/lzma/C/LzmaEnc.c: GetOptimum 1219-1968 for
// levels 5-9, GetOptimumFast 1970-2098 for levels 1-4, LzmaEnc_CodeOneBlock 2383-2680), so WHAT is decided
// is fixed: which candidate edges exist at a position, their prices, which of two equal prices wins, how far
// ahead the search may look, when the price tables are refreshed.  HOW it is computed is this file's own:
//
//   * The search is a shortest-path relaxation over a window of "arrival" nodes kept as parallel arrays
//     (structure of arrays): cost[], via[] (one packed 64-bit edge descriptor), state[], reps[] (one 128-bit
//     quadruple).  Relaxing an edge is branch-free -- compare, two conditional moves -- because whether a
//     candidate beats the incumbent is a coin toss the branch predictor loses.
//   * "Every length from a to b of one repeat / one match distance" is ONE span relaxation: the length
//     prices are contiguous in the length (lzma_model.h), the nodes are contiguous in the arrival position,
//     so eight candidate edges are priced, compared and merged per AVX2 step.
//   * Match lists are consumed in place from the finder's packed stream (one 32-bit word per pair); nothing
//     is copied or rewritten, a list that has to wait for the next parse is just a view that stays alive.
//   * The chosen path is handed to the coder as a short queue of steps, separate from the search arrays.
//   * The range coder is branch-free per coded bit (lzma_rangecoder.h).
#include <immintrin.h>

#include <cstdlib>
#include <cstring>
#include <memory>

#include "lzma_enc.h"
#include "lzma_model.h"

// tools/parser_prof.sh builds this file with -DLZMA_PARSER_PROF: cycle laps per section of the search
#ifdef LZMA_PARSER_PROF
#include <x86intrin.h>
#include <cstdio>
static unsigned long long g_lap[16], g_lap_last;
#define LAP(k)                                 \
	do {                                   \
		unsigned long long n_ = __rdtsc(); \
		g_lap[k] += n_ - g_lap_last;       \
		g_lap_last = n_;                   \
	} while (0)
#else
#define LAP(k) ((void)0)
#endif

// The file is built twice (csrc/Makefile): for x86-64-v3 (AVX2) and for x86-64-v4 (AVX-512: the relaxation
// primitives use mask registers -- compare into a mask, masked stores -- and half the instructions); the public
// entry point picks at run time what the host CPU has.
#if defined(__AVX512F__) && defined(__AVX512VL__) && defined(__AVX512BW__) && defined(__AVX512DQ__)
#define LZMA_ISA_NS isa_v4
#define LZMA_HAVE_AVX512 1
#else
#define LZMA_ISA_NS isa_v3
#define LZMA_HAVE_AVX512 0
#endif

namespace lrzgpu {
namespace LZMA_ISA_NS {
namespace {

constexpr unsigned kWindow = 1u << 11;       // arrival nodes per search (the format's encoder looks this far)
constexpr unsigned kWindowGuard = 64;        // the search is cut when this close to the end of the window
constexpr uint32_t kLiteral = 0xFFFFFFFFu;   // step / edge code of a literal
constexpr unsigned kRepSlots = 4;            // codes 0..3 are the repeat distances, 4 + d a fresh distance d
constexpr unsigned kRefreshEvery = 64;       // matches (resp. repeat lengths) between price table refreshes

// number of equal bytes of a[] and b[] in [from, limit)  (first mismatch index, limit if none)
inline unsigned equal_until(const uint8_t *a, const uint8_t *b, unsigned from, unsigned limit)
{
	unsigned i = from;
	while (i + 8 <= limit) {
		uint64_t x, y;
		memcpy(&x, a + i, 8);
		memcpy(&y, b + i, 8);
		x ^= y;
		if (x)
			return i + ((unsigned)__builtin_ctzll(x) >> 3);
		i += 8;
	}
	while (i < limit && a[i] == b[i])
		i++;
	return i;
}
inline bool same2(const uint8_t *a, const uint8_t *b)
{
	uint16_t x, y;
	memcpy(&x, a, 2);
	memcpy(&y, b, 2);
	return x == y;
}

// One position's match list, viewed in the finder's output (pairs sorted by increasing length).
// FORMAT (lzma_mf.hip k_gather): 0 plain (len, dist-1) couples; 1 the same with the tail flag in bit 31 of len;
// 2 one word per pair: flag << 31 | (len - 2) << 25 | dist-1.
template <int FORMAT> struct PairView {
	static constexpr bool kPacked = FORMAT == 2, kFlagged = FORMAT != 0;
	const uint32_t *w = nullptr;
	unsigned count = 0; // pairs
	inline unsigned len(unsigned k) const { return kPacked ? ((w[k] >> 25) & 63) + 2 : (kFlagged ? w[2 * k] & 0x7FFFFFFFu : w[2 * k]); }
	inline uint32_t dist(unsigned k) const { return kPacked ? w[k] & 0x1FFFFFFu : w[2 * k + 1]; }
	// "after this match and one literal the next two bytes continue at the same distance" (only if kFlagged)
	inline bool tail(unsigned k) const { return (kPacked ? w[k] : w[2 * k]) >> 31; }
};

// edge descriptor: how a node was reached.  `len` bytes with distance code `code`; `pre` != 0 means the edge
// is a compound: pre == 1: one literal, then a repeat-0 match of `len`; pre >= 2: a match/repeat of pre - 1
// bytes with `code`, one literal, then a repeat-0 match of `len`.
inline uint64_t edge(unsigned len, uint32_t code, unsigned pre = 0) { return (uint64_t)code | ((uint64_t)len << 32) | ((uint64_t)pre << 48); }
inline uint32_t edge_code(uint64_t e) { return (uint32_t)e; }
inline unsigned edge_len(uint64_t e) { return (unsigned)(e >> 32) & 0xFFFF; }
inline unsigned edge_pre(uint64_t e) { return (unsigned)(e >> 48); }

// How a node is entered (the edge into it) -> coder state and repeat distances there.  sel: 0 literal, 1..4 repeat
// 0..3, 5 a fresh distance.  kinds: 0 repeat (two bytes or more), 1 short repeat (one byte at repeat 0), 2 literal
// (kind of a literal is 1 + (len == 1) = 2: its length is always 1), 3 match, 4 compound (..., literal, repeat 0).
alignas(16) constexpr uint8_t kEntryKind[8] = {1, 0, 0, 0, 0, 3, 0, 0};
alignas(16) constexpr uint8_t kEntryState[5][16] = {
	{8, 8, 8, 8, 8, 8, 8, 11, 11, 11, 11, 11},  // after_rep
	{9, 9, 9, 9, 9, 9, 9, 11, 11, 11, 11, 11},  // after_short_rep
	{0, 0, 0, 0, 1, 2, 3, 4, 5, 6, 4, 5},       // after_literal
	{7, 7, 7, 7, 7, 7, 7, 10, 10, 10, 10, 10},  // after_match
	{8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8},       // ..., one literal, then a repeat
};
#define LRZ_SHUF4(a, b, c, d) {4 * a, 4 * a + 1, 4 * a + 2, 4 * a + 3, 4 * b, 4 * b + 1, 4 * b + 2, 4 * b + 3, 4 * c, 4 * c + 1, 4 * c + 2, 4 * c + 3, 4 * d, 4 * d + 1, 4 * d + 2, 4 * d + 3}
alignas(16) constexpr uint8_t kEntryShuffle[6][16] = {
	LRZ_SHUF4(0, 1, 2, 3), LRZ_SHUF4(0, 1, 2, 3), LRZ_SHUF4(1, 0, 2, 3), LRZ_SHUF4(2, 0, 1, 3), LRZ_SHUF4(3, 0, 1, 2),
	{0x80, 0x80, 0x80, 0x80, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11}, // a fresh distance goes in front
};
#undef LRZ_SHUF4

struct Step {
	uint32_t len, code;
};

template <int FORMAT> struct BlockEncoder {
	static constexpr bool PACKED = FORMAT == 2;
	// ---- input ----------------------------------------------------------------------------------
	const uint8_t *data = nullptr;
	size_t n = 0;
	const uint8_t *counts = nullptr; // u32 entries per position (2 per pair)
	const uint32_t *words = nullptr;
	size_t fetch_pos = 0;            // next position whose list has not been taken yet
	uint64_t fetch_off = 0;          // its offset into the finder stream, in u32 entries of the unpacked form
	unsigned ahead = 0;              // positions taken from the finder beyond the coder's position

	// ---- parameters --------------------------------------------------------------------------------
	unsigned nice_len = 64; // "fast bytes": a match this long is taken without further search
	unsigned pos_mask = 3;
	unsigned dist_slots = 0;
	bool greedy = false;

	// ---- coder -----------------------------------------------------------------------------------------
	RangeEncoder rc;
	LzmaModel model;
	PriceTables prices;
	unsigned state = 0;
	uint32_t reps[kRepSlots] = {1, 1, 1, 1};
	unsigned matches_since_refresh = 0;
	int rep_lens_until_refresh = (int)kRefreshEvery;

	// ---- search window (structure of arrays) ---------------------------------------------------------------
	alignas(32) uint32_t cost[kWindow + kLenMax + 16];
	alignas(32) uint64_t via[kWindow + kLenMax + 16];
	uint8_t node_state[kWindow + kLenMax + 16];
	alignas(16) uint32_t node_reps[kWindow + kLenMax + 16][kRepSlots];

	// a list fetched for the position after the parse it ended (a match of nice_len there cuts the search)
	PairView<FORMAT> held;
	unsigned held_len = 0;
	uint32_t avail_at_fetch = 0; // bytes from the last fetched position to the end of the block

	Step queue[kWindow + 8];
	unsigned q_head = 0, q_tail = 0;

	// ---- finder stream -----------------------------------------------------------------------------------
	// The search looks at the block's bytes at every candidate distance of every position ("does the match go on
	// after one literal?"): random addresses up to a dictionary back, a cache miss each.  The lists of the
	// positions ahead are already in the stream, so those lines are requested kPrefetchAhead positions early.
	static constexpr unsigned kPrefetchAhead = 12;
	size_t pf_pos = 0;
	uint64_t pf_off = 0;
	inline void prefetch_targets()
	{
		if (FORMAT != 0)
			return; // the finder answered the question itself (tail flags): nothing to fetch
		while (pf_pos < fetch_pos + kPrefetchAhead && pf_pos < n) {
			const unsigned c = counts[pf_pos];
			PairView<FORMAT> v;
			v.w = words + pf_off;
			const uint8_t *q = data + pf_pos;
			for (unsigned k = 0; k < (c >> 1); k++)
				__builtin_prefetch(q + v.len(k) - v.dist(k));
			pf_off += c;
			pf_pos++;
		}
	}
	inline PairView<FORMAT> take()
	{
		prefetch_targets();
		ahead++;
		avail_at_fetch = (uint32_t)(n - fetch_pos);
		const unsigned c = counts[fetch_pos];
		PairView<FORMAT> v;
		v.w = words + (PACKED ? (fetch_off >> 1) : fetch_off);
		v.count = c >> 1;
		fetch_off += c;
		fetch_pos++;
		return v;
	}
	inline void skip(unsigned k)
	{
		ahead += k;
		uint64_t sum = 0;
		const uint8_t *c = counts + fetch_pos;
		for (unsigned i = 0; i < k; i++)
			sum += c[i];
		fetch_off += sum;
		fetch_pos += k;
		if (pf_pos < fetch_pos) { // the skipped positions need no look-ahead any more
			pf_pos = fetch_pos;
			pf_off = fetch_off;
		}
	}
	// longest length of a list; a longest pair of exactly nice_len is extended over the bytes that follow
	// (the finder stops comparing there)
	inline unsigned longest(const PairView<FORMAT> &v) const
	{
		if (!v.count)
			return 0;
		unsigned len = v.len(v.count - 1);
		if (len != nice_len)
			return len;
		const uint32_t room = avail_at_fetch > kLenMax ? kLenMax : avail_at_fetch;
		const uint8_t *here = data + fetch_pos - 1;
		return equal_until(here, here - v.dist(v.count - 1) - 1, len, room);
	}

	// ---- relaxation primitives ------------------------------------------------------------------------------
	inline void relax(unsigned node, uint32_t c, uint64_t e)
	{
		const uint32_t old = cost[node];
		const uint64_t olde = via[node];
		const bool better = c < old;
		cost[node] = better ? c : old;
		via[node] = better ? e : olde;
	}
	// edges (len, code) for every len in [lo, hi] from node `from`: price = base + len_row[len]
#if LZMA_HAVE_AVX512
	inline void relax_span(unsigned from, unsigned lo, unsigned hi, uint32_t base, const uint32_t *len_row, uint32_t code)
	{
		const __m256i vbase = _mm256_set1_epi32((int)base);
		const __m256i iota = _mm256_setr_epi32(0, 1, 2, 3, 4, 5, 6, 7);
		const __m512i code64 = _mm512_set1_epi64((long long)code);
		for (unsigned l = lo; l <= hi; l += 8) {
			const unsigned left = hi - l + 1;
			const __mmask8 live = (__mmask8)(left >= 8 ? 0xFF : (1u << left) - 1);
			const __m256i cand = _mm256_add_epi32(vbase, _mm256_loadu_si256((const __m256i *)(len_row + l)));
			uint32_t *cp = cost + from + l;
			const __mmask8 win = _mm256_mask_cmplt_epu32_mask(live, cand, _mm256_loadu_si256((const __m256i *)cp));
			_mm256_mask_storeu_epi32(cp, win, cand);
			const __m256i lens = _mm256_add_epi32(_mm256_set1_epi32((int)l), iota);
			_mm512_mask_storeu_epi64(via + from + l, win, _mm512_or_si512(_mm512_slli_epi64(_mm512_cvtepu32_epi64(lens), 32), code64));
		}
	}
	// edges (len, code) of one fresh distance for every len in [lo, hi]: price = base + len_row[len] + the distance
	// price of the length's context, `dist4` = those four prices (lengths 2, 3, 4, 5+)
	inline void relax_pair(unsigned from, unsigned lo, unsigned hi, uint32_t base, const uint32_t *len_row, __m128i dist4, uint32_t code)
	{
		const __m256i iota = _mm256_setr_epi32(0, 1, 2, 3, 4, 5, 6, 7);
		const __m512i code64 = _mm512_set1_epi64((long long)code);
		const __m256i vbase = _mm256_add_epi32(_mm256_set1_epi32((int)base), _mm256_castsi128_si256(dist4)); // lanes 0..3 valid
		const __m256i three = _mm256_set1_epi32((int)kLenToDistStates - 1);
		for (unsigned l = lo; l <= hi; l += 8) {
			const unsigned left = hi - l + 1;
			const __mmask8 live = (__mmask8)(left >= 8 ? 0xFF : (1u << left) - 1);
			const __m256i lens = _mm256_add_epi32(_mm256_set1_epi32((int)l), iota);
			const __m256i ctx = _mm256_min_epu32(_mm256_sub_epi32(lens, _mm256_set1_epi32((int)kLenMin)), three);
			const __m256i cand = _mm256_add_epi32(_mm256_permutevar8x32_epi32(vbase, ctx), _mm256_loadu_si256((const __m256i *)(len_row + l)));
			uint32_t *cp = cost + from + l;
			const __mmask8 win = _mm256_mask_cmplt_epu32_mask(live, cand, _mm256_loadu_si256((const __m256i *)cp));
			_mm256_mask_storeu_epi32(cp, win, cand);
			_mm512_mask_storeu_epi64(via + from + l, win, _mm512_or_si512(_mm512_slli_epi64(_mm512_cvtepu32_epi64(lens), 32), code64));
		}
	}
#else
	inline void relax_span(unsigned from, unsigned lo, unsigned hi, uint32_t base, const uint32_t *len_row, uint32_t code)
	{
		const __m256i vbase = _mm256_set1_epi32((int)base);
		const __m256i iota = _mm256_setr_epi32(0, 1, 2, 3, 4, 5, 6, 7);
		const __m256i code64 = _mm256_set1_epi64x((long long)code);
		for (unsigned l = lo; l <= hi; l += 8) {
			const unsigned left = hi - l + 1; // lanes in use: min(left, 8)
			const __m256i live = _mm256_cmpgt_epi32(_mm256_set1_epi32((int)left), iota);
			const __m256i cand = _mm256_add_epi32(vbase, _mm256_loadu_si256((const __m256i *)(len_row + l)));
			uint32_t *cp = cost + from + l;
			uint64_t *vp = via + from + l;
			const __m256i old = _mm256_loadu_si256((const __m256i *)cp);
			// prices stay far below 2^31: a signed compare is an unsigned one
			const __m256i win = _mm256_and_si256(_mm256_cmpgt_epi32(old, cand), live);
			_mm256_storeu_si256((__m256i *)cp, _mm256_blendv_epi8(old, cand, win));
			const __m256i lens = _mm256_add_epi32(_mm256_set1_epi32((int)l), iota);
			const __m256i e_lo = _mm256_or_si256(_mm256_slli_epi64(_mm256_cvtepu32_epi64(_mm256_castsi256_si128(lens)), 32), code64);
			const __m256i e_hi = _mm256_or_si256(_mm256_slli_epi64(_mm256_cvtepu32_epi64(_mm256_extracti128_si256(lens, 1)), 32), code64);
			const __m256i w_lo = _mm256_cvtepi32_epi64(_mm256_castsi256_si128(win));
			const __m256i w_hi = _mm256_cvtepi32_epi64(_mm256_extracti128_si256(win, 1));
			_mm256_storeu_si256((__m256i *)vp, _mm256_blendv_epi8(_mm256_loadu_si256((const __m256i *)vp), e_lo, w_lo));
			_mm256_storeu_si256((__m256i *)(vp + 4), _mm256_blendv_epi8(_mm256_loadu_si256((const __m256i *)(vp + 4)), e_hi, w_hi));
		}
	}

	// edges (len, code) of one fresh distance for every len in [lo, hi]: price = base + len_row[len] + the distance
	// price of the length's context, `dist4` = those four prices (lengths 2, 3, 4, 5+)
	inline void relax_pair(unsigned from, unsigned lo, unsigned hi, uint32_t base, const uint32_t *len_row, __m128i dist4, uint32_t code)
	{
		const __m256i iota = _mm256_setr_epi32(0, 1, 2, 3, 4, 5, 6, 7);
		const __m256i code64 = _mm256_set1_epi64x((long long)code);
		const __m256i vbase = _mm256_add_epi32(_mm256_set1_epi32((int)base), _mm256_castsi128_si256(dist4)); // lanes 0..3 valid
		const __m256i three = _mm256_set1_epi32((int)kLenToDistStates - 1);
		for (unsigned l = lo; l <= hi; l += 8) {
			const unsigned left = hi - l + 1;
			const __m256i lens = _mm256_add_epi32(_mm256_set1_epi32((int)l), iota);
			const __m256i ctx = _mm256_min_epu32(_mm256_sub_epi32(lens, _mm256_set1_epi32((int)kLenMin)), three);
			const __m256i live = _mm256_cmpgt_epi32(_mm256_set1_epi32((int)left), iota);
			const __m256i cand = _mm256_add_epi32(_mm256_permutevar8x32_epi32(vbase, ctx), _mm256_loadu_si256((const __m256i *)(len_row + l)));
			uint32_t *cp = cost + from + l;
			uint64_t *vp = via + from + l;
			const __m256i old = _mm256_loadu_si256((const __m256i *)cp);
			const __m256i win = _mm256_and_si256(_mm256_cmpgt_epi32(old, cand), live);
			_mm256_storeu_si256((__m256i *)cp, _mm256_blendv_epi8(old, cand, win));
			const __m256i e_lo = _mm256_or_si256(_mm256_slli_epi64(_mm256_cvtepu32_epi64(_mm256_castsi256_si128(lens)), 32), code64);
			const __m256i e_hi = _mm256_or_si256(_mm256_slli_epi64(_mm256_cvtepu32_epi64(_mm256_extracti128_si256(lens, 1)), 32), code64);
			const __m256i w_lo = _mm256_cvtepi32_epi64(_mm256_castsi256_si128(win));
			const __m256i w_hi = _mm256_cvtepi32_epi64(_mm256_extracti128_si256(win, 1));
			_mm256_storeu_si256((__m256i *)vp, _mm256_blendv_epi8(_mm256_loadu_si256((const __m256i *)vp), e_lo, w_lo));
			_mm256_storeu_si256((__m256i *)(vp + 4), _mm256_blendv_epi8(_mm256_loadu_si256((const __m256i *)(vp + 4)), e_hi, w_hi));
		}
	}
#endif

	// ---- prices of the flag bits in front of a symbol ----------------------------------------------------------------
	// The model only changes between searches (the coder runs between them), and a search visits a thousand nodes in
	// at most 12 x 4 (state, position state) combinations: the sums of flag-bit prices a node needs are computed once
	// per search and combination, on first use (`flag_epoch`), and read as one line after that.
	struct FlagPrices {
		uint32_t literal;     // "a literal follows"
		uint32_t match;       // "a match with a fresh distance follows", up to its length
		uint32_t rep;         // "a repeat follows", before saying which
		uint32_t short_rep;   // one byte at repeat 0, complete
		uint32_t rep_long[4]; // repeat i of two bytes or more, up to its length
	};
	alignas(64) FlagPrices flag_price[kStates][kPosStatesMax];
	uint32_t flag_tag[kStates][kPosStatesMax];
	uint32_t flag_epoch = 0;
	__attribute__((always_inline)) inline const FlagPrices &flags(unsigned st, unsigned ps)
	{
		if (__builtin_expect(flag_tag[st][ps] != flag_epoch, 0))
			fill_flags(st, ps);
		return flag_price[st][ps];
	}
	__attribute__((noinline)) void fill_flags(unsigned st, unsigned ps)
	{
		FlagPrices &f = flag_price[st][ps];
		flag_tag[st][ps] = flag_epoch;
		const Prob m = model.is_match[st][ps];
		const uint32_t as_match = prices.bit.one(m), as_rep = as_match + prices.bit.one(model.is_rep[st]);
		f.literal = prices.bit.zero(m);
		f.match = as_match + prices.bit.zero(model.is_rep[st]);
		f.rep = as_rep;
		const uint32_t r0 = as_rep + prices.bit.zero(model.is_rep0[st]), rx = as_rep + prices.bit.one(model.is_rep0[st]);
		f.short_rep = r0 + prices.bit.zero(model.is_rep0_long[st][ps]);
		f.rep_long[0] = r0 + prices.bit.one(model.is_rep0_long[st][ps]);
		f.rep_long[1] = rx + prices.bit.zero(model.is_rep1[st]);
		const uint32_t ry = rx + prices.bit.one(model.is_rep1[st]);
		f.rep_long[2] = ry + prices.bit.zero(model.is_rep2[st]);
		f.rep_long[3] = ry + prices.bit.one(model.is_rep2[st]);
	}
	inline uint32_t price_literal(uint32_t pos, unsigned st, const uint8_t *p, unsigned match_byte) const
	{
		const Prob *ctx = model.literal_context(pos, p[-1]);
		return last_was_literal(st) ? prices.literal(ctx, p[0]) : prices.literal_matched(ctx, p[0], match_byte);
	}

	// ---- the parse -----------------------------------------------------------------------------------------
	inline unsigned single(uint32_t code, Step *first)
	{
		first->len = 1;
		first->code = code;
		return 1;
	}

	// Compound edge "X of x_len bytes, one literal, repeat-0": X ends at here + x_len with distance `dist`;
	// if at least two bytes after the literal continue at that distance, the whole thing is one candidate.
	// `price_x` = price up to and including X.  Returns the node it reaches (0 if none).
	// `known`: -1 = look at the bytes; 0 / 1 = the finder's tail flag for exactly this (x_len, dist).
	inline unsigned try_literal_then_rep0(unsigned cur, const uint8_t *here, uint32_t position, unsigned x_len, uint32_t dist,
					       unsigned st_after_x, uint32_t price_x, uint32_t room, uint32_t code, int known = -1)
	{
		const uint8_t *there = here - dist;
		unsigned tail = x_len + 1; // the literal's index
		unsigned limit = tail + nice_len;
		if (limit > room)
			limit = room;
		tail += 2;
		if (tail > limit || (known >= 0 ? !known : !same2(here + tail - 2, there + tail - 2)))
			return 0;
		const unsigned end = equal_until(here, there, tail, limit);
		const unsigned rep_len = end - x_len - 1;
		const unsigned ps_lit = (position + x_len) & pos_mask;
		uint32_t pr = price_x + flags(st_after_x, ps_lit).literal +
			      prices.literal_matched(model.literal_context(position + x_len, here[x_len - 1]), here[x_len], there[x_len]);
		const unsigned st_lit = after_literal(st_after_x), ps_rep = (ps_lit + 1) & pos_mask;
		pr += flags(st_lit, ps_rep).rep_long[0] + prices.rep_len.row[ps_rep][rep_len];
		const unsigned node = cur + end;
		relax(node, pr, edge(rep_len, code, x_len + 1));
		return node;
	}

	unsigned plan(uint32_t position, Step *first)
	{
		// ---- the root: what can start at the coder's position -------------------------------------------
		PairView<FORMAT> list;
		unsigned main_len;
		if (ahead == 0) {
			list = take();
			main_len = longest(list);
		} else {
			list = held;
			main_len = held_len;
		}
		uint32_t room = avail_at_fetch;
		if (room < 2)
			return single(kLiteral, first);
		if (room > kLenMax)
			room = kLenMax;
		const uint8_t *here = data + fetch_pos - 1;

		unsigned rep_len[kRepSlots] = {0, 0, 0, 0};
		unsigned best_rep = 0;
		for (unsigned i = 0; i < kRepSlots; i++) {
			const uint8_t *there = here - reps[i];
			if (!same2(here, there))
				continue;
			const unsigned len = equal_until(here, there, 2, room);
			rep_len[i] = len;
			if (len > rep_len[best_rep])
				best_rep = i;
			if (len == kLenMax)
				break;
		}
		if (rep_len[best_rep] >= nice_len) {
			first->len = rep_len[best_rep];
			first->code = best_rep;
			skip(first->len - 1);
			return first->len;
		}
		if (main_len >= nice_len) {
			first->len = main_len;
			first->code = list.dist(list.count - 1) + kRepSlots;
			skip(main_len - 1);
			return main_len;
		}
		const unsigned cur_byte = here[0], match_byte = here[-(ptrdiff_t)reps[0]];
		unsigned frontier = rep_len[best_rep] > main_len ? rep_len[best_rep] : main_len; // furthest node with an edge into it
		if (frontier < 2 && cur_byte != match_byte)
			return single(kLiteral, first);

		const unsigned ps0 = position & pos_mask;
		node_state[0] = (uint8_t)state;
		memcpy(node_reps[0], reps, sizeof(reps));
		flag_epoch++;
		{
			const FlagPrices &f = flags(state, ps0);
			cost[1] = f.literal + price_literal(position, state, here, match_byte);
			via[1] = edge(1, kLiteral);
			if (match_byte == cur_byte && rep_len[0] == 0) {
				relax(1, f.short_rep, edge(1, 0));
				if (frontier < 2) {
					const uint32_t code = edge_code(via[1]);
					cost[1] = kPriceInfinite;
					return single(code, first);
				}
			}
			for (unsigned i = 0; i < kRepSlots; i++)
				if (rep_len[i] >= 2)
					relax_span(0, 2, rep_len[i], f.rep_long[i], prices.rep_len.row[ps0], i);
			// fresh distances: lengths up to rep_len[0] are never cheaper than the repeat, they are not tried
			unsigned lo = rep_len[0] + 1;
			if (lo < 2)
				lo = 2;
			if (lo <= main_len)
				relax_matches(0, list, list.count, lo, main_len, f.match, ps0, nullptr, 0, 0, 0, 0, &frontier);
		}

		// ---- expand node after node until the frontier is reached -----------------------------------------
		unsigned cur = 0;
		for (;;) {
			if (++cur == frontier)
				break;
			if (cur >= kWindow - kWindowGuard) {
				// out of window: stop at the cheapest node at or beyond here (the later one on a tie)
				unsigned best = cur;
				uint32_t c = cost[cur];
				for (unsigned j = cur + 1; j <= frontier; j++)
					if (cost[j] <= c) {
						c = cost[j];
						best = j;
					}
				if (best != cur)
					skip(best - cur);
				cur = best;
				break;
			}
			LAP(0);
			const PairView<FORMAT> fresh = take();
			unsigned new_len = longest(fresh);
			LAP(1);
			if (new_len >= nice_len) { // a match worth taking outright starts here: the parse ends at this node
				held = fresh;
				held_len = new_len;
				break;
			}
			position++;

			// -- how this node was reached decides the coder state and the repeat distances here: looked up, not branched
			// on (whether a node is entered by a literal, a repeat or a match is as unpredictable as the data)
			const uint64_t e = via[cur];
			const unsigned elen = edge_len(e), epre = edge_pre(e);
			const uint32_t ecode = edge_code(e);
			const unsigned origin = cur - elen - epre;
			const unsigned sel = ecode + 1 < 5 ? ecode + 1 : 5; // 0 literal, 1 + i repeat i, 5 a fresh distance
			const unsigned kind = epre ? 4u : kEntryKind[sel] + (elen == 1);
			const unsigned st = kEntryState[kind][node_state[origin]];
			uint32_t r0; // repeat 0 here, straight from the register (the byte at that distance decides the node's first branch)
			{
				const __m128i o = _mm_load_si128((const __m128i *)node_reps[origin]);
				const __m128i moved = _mm_shuffle_epi8(o, _mm_load_si128((const __m128i *)kEntryShuffle[sel]));
				const uint32_t fresh_dist = sel == 5 ? ecode - kRepSlots + 1 : 0;
				const __m128i mine = _mm_or_si128(moved, _mm_cvtsi32_si128((int)fresh_dist));
				_mm_store_si128((__m128i *)node_reps[cur], mine);
				r0 = (uint32_t)_mm_cvtsi128_si32(mine);
			}
			node_state[cur] = (uint8_t)st;
			const uint32_t *r = node_reps[cur];
			LAP(2);

			here = data + fetch_pos - 1;
			const unsigned cb = here[0], mb = here[-(ptrdiff_t)r0];
			const unsigned ps = position & pos_mask;
			const uint32_t here_cost = cost[cur];
			const FlagPrices &f = flags(st, ps);
			const uint32_t as_rep = here_cost + f.rep;

			// -- literal.  Not priced when the next node already has an edge and the byte equals the repeat-0
			// byte (a repeat will cover it), nor when the flag bit alone already costs more than the incumbent.
			uint32_t lit_cost = here_cost + f.literal;
			bool lit_priced = false, lit_won = false;
			if (!((cost[cur + 1] < kPriceInfinite && mb == cb) || lit_cost > cost[cur + 1])) {
				lit_cost += price_literal(position, st, here, mb);
				lit_priced = true;
				if (lit_cost < cost[cur + 1]) {
					cost[cur + 1] = lit_cost;
					via[cur + 1] = edge(1, kLiteral);
					lit_won = true;
				}
			}
			// -- short repeat (one byte at repeat-0), only straight after a literal
			if (last_was_literal(st) && mb == cb && as_rep < cost[cur + 1]) {
				const uint64_t inc = via[cur + 1];
				if (edge_len(inc) < 2 || edge_code(inc) != 0) {
					const uint32_t c = here_cost + f.short_rep;
					if (c < cost[cur + 1]) {
						cost[cur + 1] = c;
						via[cur + 1] = edge(1, 0);
						lit_won = false;
					}
				}
			}

			LAP(3);
			uint32_t room_full = avail_at_fetch;
			if (room_full > kWindow - 1 - cur)
				room_full = kWindow - 1 - cur;
			if (room_full < 2)
				continue;
			const unsigned room_nice = room_full <= nice_len ? room_full : nice_len;

			// -- literal, then repeat-0
			if (!lit_won && lit_priced && mb != cb && room_full > 2) {
				const uint8_t *there = here - r0;
				if (same2(here + 1, there + 1)) {
					unsigned limit = nice_len + 1;
					if (limit > room_full)
						limit = room_full;
					const unsigned end = equal_until(here, there, 3, limit);
					const unsigned st2 = after_literal(st), ps2 = (position + 1) & pos_mask;
					const unsigned node = cur + end;
					if (frontier < node)
						frontier = node;
					relax(node, lit_cost + flags(st2, ps2).rep_long[0] + prices.rep_len.row[ps2][end - 1], edge(end - 1, 0, 1));
				}
			}

			LAP(4);
			// -- repeats
			unsigned match_from = 2; // shortest fresh-distance length worth pricing
			// which of the four continue for two bytes: found for all four at once, then only those are visited
			unsigned rep_hits;
			{
				uint16_t h, t0, t1, t2, t3;
				memcpy(&h, here, 2);
				memcpy(&t0, here - r0, 2);
				memcpy(&t1, here - r[1], 2);
				memcpy(&t2, here - r[2], 2);
				memcpy(&t3, here - r[3], 2);
				rep_hits = (unsigned)(t0 == h) | (unsigned)(t1 == h) << 1 | (unsigned)(t2 == h) << 2 | (unsigned)(t3 == h) << 3;
			}
			for (; rep_hits; rep_hits &= rep_hits - 1) {
				const unsigned i = (unsigned)__builtin_ctz(rep_hits);
				const uint8_t *there = here - r[i];
				const unsigned len = equal_until(here, there, 2, room_nice);
				if (frontier < cur + len)
					frontier = cur + len;
				const uint32_t base = here_cost + f.rep_long[i];
				relax_span(cur, 2, len, base, prices.rep_len.row[ps], i);
				if (i == 0)
					match_from = len + 1;
				const unsigned node = try_literal_then_rep0(cur, here, position, len, r[i], after_rep(st), base + prices.rep_len.row[ps][len], room_full, i);
				if (frontier < node)
					frontier = node;
			}

			LAP(5);
			// -- fresh distances
			unsigned pairs = fresh.count;
			if (new_len > room_nice) { // the list may reach past what can still be used: clip it
				new_len = room_nice;
				pairs = 0;
				while (new_len > fresh.len(pairs))
					pairs++;
				pairs++; // the first pair at least that long stands for new_len
			}
			if (new_len >= match_from) {
				if (frontier < cur + new_len)
					frontier = cur + new_len;
				relax_matches(cur, fresh, pairs, match_from, new_len, here_cost + f.match, ps, here, position, st, room_full, 1, &frontier);
			}
		}

		// ---- the window is clean again for the next parse; walk the winning path back ----------------------
		LAP(0);
		for (unsigned k = 1; k <= frontier; k++)
			cost[k] = kPriceInfinite;
		{
			const unsigned r_ = trace_back(cur, first);
			LAP(7);
			return r_;
		}
	}

	// edges of fresh distances from node `cur`: for every length lo..hi the nearest distance that reaches it
	// (pair k covers the lengths above pair k-1's up to its own; the last usable pair is clipped to hi),
	// plus, with `compound`, the "match, literal, repeat-0" edge at the full length of every pair.
	inline void relax_matches(unsigned cur, const PairView<FORMAT> &list, unsigned pairs, unsigned lo, unsigned hi, uint32_t base,
				  unsigned ps, const uint8_t *here, uint32_t position, unsigned st, uint32_t room_full, int compound,
				  unsigned *frontier)
	{
		const uint32_t *len_row = prices.match_len.row[ps];
#if LZMA_HAVE_AVX512
		// The common shape -- at most four pairs, at most sixteen lengths, the list not clipped, no tail flag up -- as
		// straight-line code: every length lo..hi is one lane; pair k takes the lanes up to its length that no
		// shorter pair has taken (a pair beyond the list's end repeats the last one and finds no lane left), the
		// lanes' distance prices and edge codes are merged under those masks, then ONE compare-and-store prices all
		// of it.  The loop below does the same pair by pair; its trip count is a mispredicted branch per position.
		if (PACKED && pairs <= 4 && hi - lo < 16 && hi == list.len(pairs - 1)) {
			// the (at most four) words of the list as one vector; lanes beyond the list are zero and inert: length 2, no
			// lane left for them (the last real pair has taken everything up to hi)
			const __m128i wv = _mm_maskz_loadu_epi32((__mmask8)((1u << pairs) - 1), list.w);
			if (!compound || !_mm_movemask_ps(_mm_castsi128_ps(wv))) {
				const __m512i lens = _mm512_add_epi32(_mm512_set1_epi32((int)lo), _mm512_setr_epi32(0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15));
				const __mmask16 live = _mm512_cmple_epu32_mask(lens, _mm512_set1_epi32((int)hi));
				const __m512i ctx = _mm512_min_epu32(_mm512_sub_epi32(lens, _mm512_set1_epi32((int)kLenMin)), _mm512_set1_epi32((int)kLenToDistStates - 1));
				// per pair, for all four at once: distance, length, edge code, the row of its distance prices (a near
				// distance has its own row; a far one the row of its slot = 2 * floor(log2 d) + the bit below the top
				// one, and the price of its low four bits on top)
				const __m128i dv = _mm_and_si128(wv, _mm_set1_epi32(0x1FFFFFF));
				const __m128i plenv = _mm_add_epi32(_mm_and_si128(_mm_srli_epi32(wv, 25), _mm_set1_epi32(63)), _mm_set1_epi32((int)kLenMin));
				const __mmask8 far = _mm_cmpge_epu32_mask(dv, _mm_set1_epi32((int)kNearDistances));
				const __m128i top = _mm_sub_epi32(_mm_set1_epi32(31), _mm_lzcnt_epi32(dv));
				const __m128i below = _mm_and_si128(_mm_srlv_epi32(dv, _mm_sub_epi32(top, _mm_set1_epi32(1))), _mm_set1_epi32(1));
				const __m128i far_row = _mm_add_epi32(_mm_add_epi32(_mm_slli_epi32(top, 1), below), _mm_set1_epi32((int)kNearDistances));
				const __m128i row_off = _mm_slli_epi32(_mm_mask_mov_epi32(dv, far, far_row), 4); // bytes into dist_rows
				const __m512i plen16 = _mm512_castsi128_si512(plenv), code16 = _mm512_castsi128_si512(_mm_add_epi32(dv, _mm_set1_epi32((int)kRepSlots)));
				const __m512i low16 = _mm512_maskz_permutexvar_epi32((__mmask16)far, _mm512_castsi128_si512(_mm_and_si128(dv, _mm_set1_epi32((int)kAlignSize - 1))),
										  _mm512_load_si512((const void *)prices.align));
				// pair j takes the lanes up to its length that pair j - 1 has not taken (lengths increase along the list):
				// four independent compares, not a chain through "what is still open"; the three per-lane quantities are
				// picked per pair under those masks and merged pairwise (short dependency chains: the node's edges
				// cannot be priced before this is through, and the next node cannot retire before this one)
				const __m128i row_off_x = row_off;
				const char *rows = (const char *)prices.dist_rows;
				const __mmask16 le0 = _mm512_mask_cmple_epu32_mask(live, lens, _mm512_permutexvar_epi32(_mm512_set1_epi32(0), plen16));
				const __mmask16 le1 = _mm512_mask_cmple_epu32_mask(live, lens, _mm512_permutexvar_epi32(_mm512_set1_epi32(1), plen16));
				const __mmask16 le2 = _mm512_mask_cmple_epu32_mask(live, lens, _mm512_permutexvar_epi32(_mm512_set1_epi32(2), plen16));
				const __mmask16 le3 = _mm512_mask_cmple_epu32_mask(live, lens, _mm512_permutexvar_epi32(_mm512_set1_epi32(3), plen16));
				const __mmask16 m0 = le0, m1 = (__mmask16)(le1 & ~le0), m2 = (__mmask16)(le2 & ~(le1 | le0)), m3 = (__mmask16)(le3 & ~(le2 | le1 | le0));
#define LRZ_ROW(j) _mm512_castsi128_si512(_mm_load_si128((const __m128i *)(rows + (uint32_t)_mm_extract_epi32(row_off_x, j))))
				const __m512i dist_price = _mm512_or_si512(
					_mm512_or_si512(_mm512_maskz_permutexvar_epi32(m0, ctx, LRZ_ROW(0)), _mm512_maskz_permutexvar_epi32(m1, ctx, LRZ_ROW(1))),
					_mm512_or_si512(_mm512_maskz_permutexvar_epi32(m2, ctx, LRZ_ROW(2)), _mm512_maskz_permutexvar_epi32(m3, ctx, LRZ_ROW(3))));
#undef LRZ_ROW
				// the lane's pair: 0..3 (a lane no pair takes is not live)
				const __m512i which = _mm512_mask_set1_epi32(_mm512_mask_set1_epi32(_mm512_maskz_set1_epi32(m1, 1), m2, 2), m3, 3);
				const __m512i codes = _mm512_permutexvar_epi32(which, code16);
				const __m512i low_price = _mm512_permutexvar_epi32(which, low16);
				const __m512i cand = _mm512_add_epi32(_mm512_add_epi32(_mm512_set1_epi32((int)base), dist_price), _mm512_add_epi32(low_price, _mm512_maskz_loadu_epi32(live, len_row + lo)));
				uint32_t *cp = cost + cur + lo;
				const __mmask16 win = _mm512_mask_cmplt_epu32_mask(live, cand, _mm512_maskz_loadu_epi32(live, cp));
				_mm512_mask_storeu_epi32(cp, win, cand);
				uint64_t *vp = via + cur + lo;
				const __m512i e_lo = _mm512_or_si512(_mm512_slli_epi64(_mm512_cvtepu32_epi64(_mm512_castsi512_si256(lens)), 32), _mm512_cvtepu32_epi64(_mm512_castsi512_si256(codes)));
				const __m512i e_hi = _mm512_or_si512(_mm512_slli_epi64(_mm512_cvtepu32_epi64(_mm512_extracti64x4_epi64(lens, 1)), 32), _mm512_cvtepu32_epi64(_mm512_extracti64x4_epi64(codes, 1)));
				_mm512_mask_storeu_epi64(vp, (__mmask8)win, e_lo);
				_mm512_mask_storeu_epi64(vp + 8, (__mmask8)(win >> 8), e_hi);
				return;
			}
		}
#endif
		unsigned k = 0;
		while (lo > list.len(k) && k + 1 < pairs)
			k++;
		unsigned len = lo;
		for (; k < pairs; k++) {
			unsigned top = list.len(k);
			if (k + 1 == pairs || top > hi)
				top = hi;
			const uint32_t d = list.dist(k);
			const uint32_t code = d + kRepSlots;
			// the distance price in the four length contexts: near distances from one table, far ones slot + align
			alignas(16) uint32_t dp[4];
			__m128i dist4;
			if (d < kNearDistances)
				dist4 = _mm_load_si128((const __m128i *)prices.near_dist[d]);
			else
				dist4 = _mm_add_epi32(_mm_load_si128((const __m128i *)prices.slot[dist_slot(d)]), _mm_set1_epi32((int)prices.align[d & (kAlignSize - 1)]));
			// a node reached through this pair reads the byte that follows the match source (its repeat-0 byte)
			// when it is expanded, `top` positions from now: have the line on its way
			__builtin_prefetch(data + fetch_pos - 1 + top - d - 1);
			if (len <= top) {
				relax_pair(cur, len, top, base, len_row, dist4, code);
				len = top + 1;
			}
			// the flag speaks for the pair's own length (a clipped last pair has to look at the bytes) and it says "no"
			// for 999 pairs of 1000: nothing of the compound edge is computed for those
			if (compound && !(PairView<FORMAT>::kFlagged && top == list.len(k) && !list.tail(k))) {
				_mm_store_si128((__m128i *)dp, dist4);
				const uint32_t price_x = base + len_row[top] + dp[len_dist_state(top)];
				const int known = PairView<FORMAT>::kFlagged && top == list.len(k) ? 1 : -1;
				const unsigned node = try_literal_then_rep0(cur, here, position, top, d + 1, after_match(st), price_x, room_full, code, known);
				if (*frontier < node)
					*frontier = node;
			}
			if (top == hi)
				break;
		}
	}

	// from node `cur` back to node 0: the steps in coding order go to the queue, the first one to the caller
	unsigned trace_back(unsigned cur, Step *first)
	{
		Step tmp[kWindow + 8];
		unsigned k = 0;
		while (cur) {
			const uint64_t e = via[cur];
			const unsigned len = edge_len(e), pre = edge_pre(e);
			const uint32_t code = edge_code(e);
			if (!pre) {
				tmp[k++] = Step{len, code};
				cur -= len;
			} else if (pre == 1) {
				tmp[k++] = Step{len, code}; // the repeat-0 (its code is 0) ...
				tmp[k++] = Step{1, kLiteral}; // ... after one literal
				cur -= len + 1;
			} else {
				tmp[k++] = Step{len, 0};
				tmp[k++] = Step{1, kLiteral};
				tmp[k++] = Step{pre - 1, code};
				cur -= len + pre;
			}
		}
		*first = tmp[--k];
		q_head = q_tail = 0;
		while (k)
			queue[q_tail++] = tmp[--k];
		return first->len;
	}

	// ---- levels 1-4: greedy with one position of look-ahead (reference GetOptimumFast) -----------------------------
	static inline bool much_nearer(uint32_t small_dist, uint32_t big_dist) { return (big_dist >> 7) > small_dist; }
	unsigned plan_greedy(Step *first)
	{
		PairView<FORMAT> list;
		unsigned main_len;
		if (ahead == 0) {
			list = take();
			main_len = longest(list);
		} else {
			list = held;
			main_len = held_len;
		}
		uint32_t room = avail_at_fetch;
		if (room < 2)
			return single(kLiteral, first);
		if (room > kLenMax)
			room = kLenMax;
		const uint8_t *here = data + fetch_pos - 1;
		unsigned rep_best = 0, rep_which = 0;
		for (unsigned i = 0; i < kRepSlots; i++) {
			const uint8_t *there = here - reps[i];
			if (!same2(here, there))
				continue;
			const unsigned len = equal_until(here, there, 2, room);
			if (len >= nice_len) {
				first->len = len;
				first->code = i;
				skip(len - 1);
				return len;
			}
			if (len > rep_best) {
				rep_which = i;
				rep_best = len;
			}
		}
		if (main_len >= nice_len) {
			first->len = main_len;
			first->code = list.dist(list.count - 1) + kRepSlots;
			skip(main_len - 1);
			return main_len;
		}
		uint32_t main_dist = 0;
		if (main_len >= 2) {
			unsigned k = list.count - 1;
			main_dist = list.dist(k);
			// one byte shorter but more than 128 times nearer is the better deal
			while (k > 0 && main_len == list.len(k - 1) + 1 && much_nearer(list.dist(k - 1), main_dist)) {
				k--;
				main_len--;
				main_dist = list.dist(k);
			}
			if (main_len == 2 && main_dist >= 0x80)
				main_len = 1;
		}
		if (rep_best >= 2 && (rep_best + 1 >= main_len || (rep_best + 2 >= main_len && main_dist >= (1u << 9)) ||
				      (rep_best + 3 >= main_len && main_dist >= (1u << 15)))) {
			first->len = rep_best;
			first->code = rep_which;
			skip(rep_best - 1);
			return rep_best;
		}
		if (main_len < 2 || room <= 2)
			return single(kLiteral, first);
		// look one position ahead: a clearly better match there makes this byte a literal
		held = take();
		held_len = longest(held);
		if (held_len >= 2) {
			const uint32_t next_dist = held.dist(held.count - 1);
			if ((held_len >= main_len && next_dist < main_dist) || (held_len == main_len + 1 && !much_nearer(main_dist, next_dist)) ||
			    held_len > main_len + 1 || (held_len + 1 >= main_len && main_len >= 3 && much_nearer(next_dist, main_dist)))
				return single(kLiteral, first);
		}
		here = data + fetch_pos - 1;
		for (unsigned i = 0; i < kRepSlots; i++) {
			const uint8_t *there = here - reps[i];
			if (!same2(here, there))
				continue;
			const unsigned limit = main_len - 1;
			if (equal_until(here, there, 2, limit) >= limit)
				return single(kLiteral, first);
		}
		first->len = main_len;
		first->code = main_dist + kRepSlots;
		if (main_len != 2)
			skip(main_len - 2);
		return main_len;
	}

	// ---- coding one step -------------------------------------------------------------------------------------------
	inline void code_step(uint32_t pos, const Step &s)
	{
		const unsigned ps = pos & pos_mask;
		Prob *flag = &model.is_match[state][ps];
		if (s.code == kLiteral) {
			rc.encode(flag, 0);
			const uint8_t *p = data + pos;
			Prob *ctx = model.literal_context(pos, p[-1]);
			if (last_was_literal(state))
				rc.encode_tree<8>(ctx, p[0]);
			else {
				unsigned offs = 0x100, sym = p[0] | 0x100u, mb = p[-(ptrdiff_t)reps[0]];
				do {
					mb <<= 1;
					Prob *pr = ctx + offs + (mb & offs) + (sym >> 8);
					const unsigned b = (sym >> 7) & 1;
					sym <<= 1;
					offs &= ~(mb ^ sym);
					rc.encode(pr, b);
				} while (sym < 0x10000);
			}
			state = after_literal(state);
			return;
		}
		rc.encode(flag, 1);
		if (s.code < kRepSlots) {
			rc.encode(&model.is_rep[state], 1);
			if (s.code == 0) {
				rc.encode(&model.is_rep0[state], 0);
				rc.encode(&model.is_rep0_long[state][ps], s.len != 1);
				if (s.len == 1) {
					state = after_short_rep(state);
					return;
				}
			} else {
				rc.encode(&model.is_rep0[state], 1);
				rc.encode(&model.is_rep1[state], s.code != 1);
				if (s.code != 1)
					rc.encode(&model.is_rep2[state], s.code - 2);
				const uint32_t d = reps[s.code];
				for (unsigned i = s.code; i > 0; i--)
					reps[i] = reps[i - 1];
				reps[0] = d;
			}
			model.rep_len.encode(rc, s.len, ps);
			--rep_lens_until_refresh;
			state = after_rep(state);
			return;
		}
		rc.encode(&model.is_rep[state], 0);
		state = after_match(state);
		model.match_len.encode(rc, s.len, ps);
		const uint32_t d = s.code - kRepSlots;
		reps[3] = reps[2];
		reps[2] = reps[1];
		reps[1] = reps[0];
		reps[0] = d + 1;
		matches_since_refresh++;
		const unsigned sl = dist_slot(d);
		rc.encode_tree<6>(model.slot[len_dist_state(s.len)], sl);
		if (d >= 4) {
			const unsigned nb = (sl >> 1) - 1;
			const uint32_t base = (uint32_t)(2 | (sl & 1)) << nb;
			if (d < kNearDistances)
				rc.encode_tree_reverse(model.near_footer + base, nb, d - base);
			else {
				rc.encode_direct((d - base) >> kAlignBits, nb - kAlignBits);
				rc.encode_tree_reverse(model.align, kAlignBits, d & (kAlignSize - 1));
			}
		}
	}

	void refresh_all()
	{
		prices.refresh_align(model);
		prices.refresh_distances(model, dist_slots);
		matches_since_refresh = 0;
		prices.match_len.refresh(model.match_len, prices.bit, 1u << model.pb, nice_len);
	}

	void setup(const LzmaParams &prm)
	{
		unsigned fb = (unsigned)prm.fb;
		if (fb < 5)
			fb = 5;
		if (fb > kLenMax)
			fb = kLenMax;
		nice_len = fb;
		greedy = prm.fast;
		unsigned i;
		for (i = 7; i < 32; i++)
			if (prm.dict_size <= (1u << i))
				break;
		dist_slots = i * 2;
		model.reset((unsigned)prm.lc, (unsigned)prm.lp, (unsigned)prm.pb);
		pos_mask = (1u << prm.pb) - 1;
		for (unsigned k = 0; k < sizeof(cost) / sizeof(cost[0]); k++)
			cost[k] = kPriceInfinite;
		memset(via, 0, sizeof(via));
		memset(flag_tag, 0, sizeof(flag_tag));
		flag_epoch = 0;
		refresh_all();
		prices.rep_len.refresh(model.rep_len, prices.bit, 1u << model.pb, nice_len);
	}

	// ---- lists that arrive in stages (lzma_enc.h StagedLists; the early start of a block) --------
	// The parse may begin while the finder has covered only a prefix of the block: positions below stage_limit have
	// lists (and the block's bytes are there up to the limit).  A search takes at most kWindow positions, a step skips
	// at most kLenMax more, and a comparison reads at most kLenMax bytes beyond the position it starts at: before a
	// search could come within kStageReach of the limit the producer is asked for more (it blocks until it has more).
	static constexpr size_t kStageReach = kWindow + 2 * kLenMax + 64;
	size_t stage_limit = 0; // 0 = everything is there
	size_t stage_have = 0;  // what the producer said last
	const MatchLists *(*stage_rest)(void *, size_t *) = nullptr;
	void *stage_ctx = nullptr;
	bool stage_failed = false;
	void more_lists()
	{
		while (stage_limit && fetch_pos + kStageReach >= stage_limit) {
			size_t valid = 0;
			const MatchLists *ml = stage_rest ? stage_rest(stage_ctx, &valid) : nullptr;
			if (!ml || ml->packed != (FORMAT == 2) || ml->tail_flags != (FORMAT != 0) || (valid < n && valid <= stage_have)) {
				stage_failed = true; // the block was withdrawn (or the producer made no progress)
				return;
			}
			if (held.w) // a list kept for the next search is a view into the old arrays: same offset in the new ones
				held.w = ml->pairs + (held.w - words);
			counts = ml->counts;
			words = ml->pairs;
			stage_have = valid;
			stage_limit = valid < n ? (valid ? valid : 1) : 0;
		}
	}

	void run()
	{
		if (n == 0) {
			rc.finish();
			return;
		}
		more_lists();
		if (stage_failed)
			return;
		// the first byte has no context and no history: always a plain literal
		(void)take();
		rc.encode(&model.is_match[0][0], 0);
		rc.encode_tree<8>(model.literal.data(), data[0]);
		ahead--;
		uint32_t pos = 1;
		if (fetch_pos < n)
			for (;;) {
				Step s;
				if (q_head != q_tail)
					s = queue[q_head++];
				else {
					if (__builtin_expect(stage_limit != 0, 0)) {
						more_lists();
						if (stage_failed)
							return;
					}
					if (greedy)
						plan_greedy(&s);
					else {
						LAP(8);
						plan(pos, &s);
					}
				}
				code_step(pos, s);
				pos += s.len;
				ahead -= s.len;
				LAP(9);
				if (ahead == 0) { // the coder has caught up with the finder: the only moment tables may change
					if (!greedy && matches_since_refresh >= kRefreshEvery)
						refresh_all();
					if (!greedy && rep_lens_until_refresh <= 0) {
						rep_lens_until_refresh = (int)kRefreshEvery;
						prices.rep_len.refresh(model.rep_len, prices.bit, 1u << model.pb, nice_len);
					}
					LAP(10);
					if (fetch_pos == n || rc.overflow)
						break; // (an overflow ends in LZ_ERROR_OUTPUT_EOF whatever follows)
				}
			}
		rc.finish();
	}
};

template <int FORMAT>
int encode_staged_with(const LzmaParams &prm, const uint8_t *src, size_t n, const StagedLists &sl, uint8_t *dest, size_t dest_cap, size_t *dest_len)
{
	std::unique_ptr<BlockEncoder<FORMAT>> e(new (std::nothrow) BlockEncoder<FORMAT>());
	if (!e)
		return LZ_ERROR_MEM;
	e->data = src;
	e->n = n;
	e->counts = sl.early.counts;
	e->words = sl.early.pairs;
	e->stage_limit = sl.early_positions < n ? (sl.early_positions ? sl.early_positions : 1) : 0; // (1: ask at once)
	e->stage_have = sl.early_positions;
	e->stage_rest = sl.rest;
	e->stage_ctx = sl.ctx;
	e->rc.out = dest;
	e->rc.cap = dest_cap;
	e->setup(prm);
	e->run();
	if (e->stage_failed)
		return LZ_ERROR_PARAM;
	if (e->rc.overflow) {
		*dest_len = dest_cap;
		return LZ_ERROR_OUTPUT_EOF;
	}
	*dest_len = e->rc.len;
	return LZ_OK;
}

template <int FORMAT>
int encode_with(const LzmaParams &prm, const uint8_t *src, size_t n, const MatchLists &ml, uint8_t *dest, size_t dest_cap, size_t *dest_len)
{
	std::unique_ptr<BlockEncoder<FORMAT>> e(new (std::nothrow) BlockEncoder<FORMAT>());
	if (!e)
		return LZ_ERROR_MEM;
	e->data = src;
	e->n = n;
	e->counts = ml.counts;
	e->words = ml.pairs;
	e->rc.out = dest;
	e->rc.cap = dest_cap;
	e->setup(prm);
#ifdef LZMA_PARSER_PROF
	g_lap_last = __rdtsc();
#endif
	e->run();
#ifdef LZMA_PARSER_PROF
	{
		static const char *nm[] = {"matches + loop ends", "take list", "enter node", "literal + short rep", "literal, rep0", "repeats", "-", "reset + trace back", "root (plan start)", "code step", "table refresh"};
		unsigned long long tot = 0;
		for (int k = 0; k < 11; k++)
			tot += g_lap[k];
		for (int k = 0; k < 11; k++)
			fprintf(stderr, "%-22s %6.2f%%  %.1f cyc/byte\n", nm[k], 100.0 * g_lap[k] / tot, (double)g_lap[k] / n);
		fprintf(stderr, "total %.1f cyc/byte (each lap costs ~25-30 cycles itself)\n", (double)tot / n);
	}
#endif
	if (e->rc.overflow) {
		*dest_len = dest_cap;
		return LZ_ERROR_OUTPUT_EOF;
	}
	*dest_len = e->rc.len;
	return LZ_OK;
}


} // namespace

int encode_block(const LzmaParams &prm, const uint8_t *src, size_t n, const MatchLists &ml, uint8_t *dest, size_t dest_cap, size_t *dest_len)
{
	if (ml.packed)
		return encode_with<2>(prm, src, n, ml, dest, dest_cap, dest_len);
	if (ml.tail_flags)
		return encode_with<1>(prm, src, n, ml, dest, dest_cap, dest_len);
	return encode_with<0>(prm, src, n, ml, dest, dest_cap, dest_len);
}
int encode_block_staged(const LzmaParams &prm, const uint8_t *src, size_t n, const StagedLists &sl, uint8_t *dest, size_t dest_cap, size_t *dest_len)
{
	if (sl.early.packed)
		return encode_staged_with<2>(prm, src, n, sl, dest, dest_cap, dest_len);
	if (sl.early.tail_flags)
		return encode_staged_with<1>(prm, src, n, sl, dest, dest_cap, dest_len);
	return encode_staged_with<0>(prm, src, n, sl, dest, dest_cap, dest_len);
}

} // namespace LZMA_ISA_NS
} // namespace lrzgpu

#if !LZMA_HAVE_AVX512 // the ISA-independent rest lives in the baseline build only
namespace lrzgpu {
namespace isa_v4 {
int encode_block(const LzmaParams &prm, const uint8_t *src, size_t n, const MatchLists &ml, uint8_t *dest, size_t dest_cap, size_t *dest_len);
int encode_block_staged(const LzmaParams &prm, const uint8_t *src, size_t n, const StagedLists &sl, uint8_t *dest, size_t dest_cap, size_t *dest_len);
}

// hash masks the reference derives for a block (LzFind.c:347-373, 432-442): smallest 2^k - 1 covering
// min(dictionary, expected size), halved, at most 2^24 - 1 worth of bits for 4 hash bytes, at least 16 bits
static uint32_t hash_mask_common(uint32_t dict_size, uint64_t expected_size, uint32_t always_set)
{
	uint32_t best = 0xFFFFFFFFu;
	const uint64_t sizes[2] = {dict_size, expected_size < dict_size ? expected_size : dict_size};
	for (uint64_t sz : sizes) {
		uint32_t hs = (uint32_t)sz;
		if (hs)
			hs--;
		hs |= hs >> 1;
		hs |= hs >> 2;
		hs |= hs >> 4;
		hs |= hs >> 8;
		hs >>= 1;
		if (hs >= (1u << 24))
			hs >>= 1;
		hs |= 0xFFFF | always_set;
		if (hs < best)
			best = hs;
	}
	return best;
}
uint32_t lzma_hash_mask(uint32_t dict_size, uint64_t expected_size) { return hash_mask_common(dict_size, expected_size, 0); }
// the 5-byte hash of the HC5 finder (levels 1-4) always keeps its low 18 bits (kLzHash_CrcShift_2)
uint32_t lzma_hash_mask5(uint32_t dict_size, uint64_t expected_size) { return hash_mask_common(dict_size, expected_size, (256u << 10) - 1); }

// 5-byte properties: lc/lp/pb packed, then the dictionary rounded up to 2^k or 3 * 2^(k-1) below 2 MiB,
// to a whole MiB above (LzmaEnc_WriteProperties, LzmaEnc.c:3037-3070)
void lzma_write_props(const LzmaParams &prm, uint8_t props[5])
{
	const uint32_t dict = prm.dict_size;
	uint32_t v;
	props[0] = (uint8_t)((prm.pb * 5 + prm.lp) * 9 + prm.lc);
	if (dict >= (1u << 21)) {
		const uint32_t mib = (1u << 20) - 1;
		v = (dict + mib) & ~mib;
		if (v < dict)
			v = dict;
	} else {
		v = 1u << 12; // 2^12, 3 * 2^11 ... : i = 22, 23, ... in (2 + (i & 1)) << (i >> 1)
		for (unsigned i = 11 * 2; (v = (uint32_t)(2 + (i & 1)) << (i >> 1)) < dict; i++) {
		}
	}
	for (int k = 0; k < 4; k++)
		props[1 + k] = (uint8_t)(v >> (8 * k));
}

int lzma_encode_block(const LzmaParams &prm, const uint8_t *src, size_t n, const MatchLists &ml, uint8_t *dest, size_t dest_cap,
		      size_t *dest_len)
{
	if (prm.lc > 8 || prm.lp > 4 || prm.pb > 4 || prm.lc < 0 || prm.lp < 0 || prm.pb < 0)
		return LZ_ERROR_PARAM;
	if ((prm.level < 5) != prm.fast) // algo 0 <=> levels 1-4 here (LzmaEncProps_Normalize)
		return LZ_ERROR_PARAM;
	if (n >= 0xFFFFFFFFu)
		return LZ_ERROR_PARAM;
	static const bool v4 = __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512vl") && __builtin_cpu_supports("avx512bw") &&
			       __builtin_cpu_supports("avx512dq") && !getenv("LRZGPU_NO_AVX512");
	return v4 ? isa_v4::encode_block(prm, src, n, ml, dest, dest_cap, dest_len) : isa_v3::encode_block(prm, src, n, ml, dest, dest_cap, dest_len);
}

int lzma_encode_block_staged(const LzmaParams &prm, const uint8_t *src, size_t n, const StagedLists &sl, uint8_t *dest, size_t dest_cap,
			     size_t *dest_len)
{
	if (prm.lc > 8 || prm.lp > 4 || prm.pb > 4 || prm.lc < 0 || prm.lp < 0 || prm.pb < 0 || (prm.level < 5) != prm.fast || n >= 0xFFFFFFFFu ||
	    !sl.rest)
		return LZ_ERROR_PARAM;
	static const bool v4 = __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512vl") && __builtin_cpu_supports("avx512bw") &&
			       __builtin_cpu_supports("avx512dq") && !getenv("LRZGPU_NO_AVX512");
	return v4 ? isa_v4::encode_block_staged(prm, src, n, sl, dest, dest_cap, dest_len)
		  : isa_v3::encode_block_staged(prm, src, n, sl, dest, dest_cap, dest_len);
}

} // namespace lrzgpu
#endif
