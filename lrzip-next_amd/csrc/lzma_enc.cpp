// lzma_enc.cpp -- host LZMA optimal parser + range coder fed by GPU match lists.
//
// Behavioural contract (bit-exact): reference src/lzma/C/LzmaEnc.c, algo=1 (levels 5-9):
//   range coder            LzmaEnc.c:631-759      -> RangeCoder
//   literal/len coders     LzmaEnc.c:789-826, 928-961
//   price tables           LzmaEnc.c:830-897, 963-1065, 2202-2320
//   ReadMatchDistances     LzmaEnc.c:1079-1123    -> Encoder::read_matches
//   GetOptimum / Backward  LzmaEnc.c:1167-1968    -> Encoder::optimum / backward
//   LzmaEnc_CodeOneBlock   LzmaEnc.c:2383-2680    -> Encoder::run
// The match finder is not here: lists come from lzma_mf.hip (or, in CPU-only unit tests, from
// any other producer of the same lists).
#include "lzma_enc.h"

#include <cstring>
#include <memory>

namespace lrzgpu {
namespace {

typedef uint16_t Prob;

constexpr unsigned kBitModelBits = 11;
constexpr unsigned kBitModelTotal = 1u << kBitModelBits;
constexpr unsigned kMoveBits = 5;
constexpr Prob kProbInit = kBitModelTotal >> 1;
constexpr unsigned kMoveReducingBits = 4;
constexpr unsigned kBitPriceShift = 4;
constexpr uint32_t kTopValue = 1u << 24;

constexpr unsigned kNumReps = 4;
constexpr unsigned kNumOpts = 1u << 11;
constexpr unsigned kNumLenToPosStates = 4;
constexpr unsigned kNumPosSlotBits = 6;
constexpr unsigned kDistTableSizeMax = 64;
constexpr unsigned kNumAlignBits = 4;
constexpr unsigned kAlignTableSize = 1u << kNumAlignBits;
constexpr unsigned kAlignMask = kAlignTableSize - 1;
constexpr unsigned kStartPosModelIndex = 4;
constexpr unsigned kEndPosModelIndex = 14;
constexpr unsigned kNumFullDistances = 1u << (kEndPosModelIndex >> 1);
constexpr unsigned kLenLowBits = 3;
constexpr unsigned kLenLowSymbols = 1u << kLenLowBits;
constexpr unsigned kLenHighBits = 8;
constexpr unsigned kLenHighSymbols = 1u << kLenHighBits;
constexpr unsigned kLenSymbolsTotal = kLenLowSymbols * 2 + kLenHighSymbols;
constexpr unsigned kMatchLenMin = 2;
constexpr unsigned kMatchLenMax = kMatchLenMin + kLenSymbolsTotal - 1; // 273
constexpr unsigned kNumStates = 12;
constexpr unsigned kPbStatesMax = 16;
constexpr uint32_t kInfinityPrice = 1u << 30;
constexpr int kRepLenCount = 64;
constexpr uint32_t kMarkLit = 0xFFFFFFFFu;

const uint8_t kLitNext[kNumStates] = {0, 0, 0, 0, 1, 2, 3, 4, 5, 6, 4, 5};
const uint8_t kMatchNext[kNumStates] = {7, 7, 7, 7, 7, 7, 7, 10, 10, 10, 10, 10};
const uint8_t kRepNext[kNumStates] = {8, 8, 8, 8, 8, 8, 8, 11, 11, 11, 11, 11};
const uint8_t kShortRepNext[kNumStates] = {9, 9, 9, 9, 9, 9, 9, 11, 11, 11, 11, 11};
constexpr unsigned kStateLitAfterMatch = 4, kStateLitAfterRep = 5, kStateMatchAfterLit = 7, kStateRepAfterLit = 8;

inline bool is_lit_state(unsigned s) { return s < 7; }
// first index in [from, limit) where a[] and b[] differ (limit if none): eight bytes per step
inline unsigned mismatch_from(const uint8_t *a, const uint8_t *b, unsigned from, unsigned limit)
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
inline unsigned len_to_pos_state(unsigned len) { return len < kNumLenToPosStates + 1 ? len - 2 : kNumLenToPosStates - 1; }

inline unsigned pos_slot(uint32_t d)
{
	if (d < 2)
		return d;
	unsigned i = 31 - (unsigned)__builtin_clz(d);
	return (i << 1) + ((d >> (i - 1)) & 1);
}

struct RangeCoder {
	uint64_t low = 0;
	uint32_t range = 0xFFFFFFFFu;
	uint32_t cache = 0;
	uint64_t cache_size = 0;
	uint8_t *out = nullptr;
	size_t cap = 0, len = 0;
	bool overflow = false;

	inline void put(uint8_t b)
	{
		if (len < cap)
			out[len] = b;
		else
			overflow = true;
		len++;
	}
	void shift_low()
	{
		uint32_t lo = (uint32_t)low;
		unsigned hi = (unsigned)(low >> 32);
		low = (uint32_t)(lo << 8);
		if (lo < 0xFF000000u || hi != 0) {
			put((uint8_t)(cache + hi));
			cache = lo >> 24;
			if (cache_size == 0)
				return;
			hi += 0xFF;
			for (;;) {
				put((uint8_t)hi);
				if (--cache_size == 0)
					return;
			}
		}
		cache_size++;
	}
	inline void norm()
	{
		if (range < kTopValue) {
			range <<= 8;
			shift_low();
		}
	}
	inline void bit(Prob *prob, uint32_t b)
	{
		uint32_t t = *prob;
		uint32_t bound = (range >> kBitModelBits) * t;
		if (b == 0) {
			range = bound;
			t += (kBitModelTotal - t) >> kMoveBits;
		} else {
			low += bound;
			range -= bound;
			t -= t >> kMoveBits;
		}
		*prob = (Prob)t;
		norm();
	}
	void direct(uint32_t value, unsigned nbits) // MSB first
	{
		while (nbits--) {
			range >>= 1;
			low += range & (0u - ((value >> nbits) & 1));
			norm();
		}
	}
	void flush()
	{
		for (int i = 0; i < 5; i++)
			shift_low();
	}
};

struct LenProbs {
	Prob low[kPbStatesMax << (kLenLowBits + 1)];
	Prob high[kLenHighSymbols];
	void init()
	{
		for (auto &p : low) p = kProbInit;
		for (auto &p : high) p = kProbInit;
	}
};

struct LenPrices {
	unsigned table_size = 0;
	uint32_t prices[kPbStatesMax][kLenSymbolsTotal];
};

struct Opt {
	uint32_t price;
	uint16_t state;
	uint16_t extra;
	uint32_t len;
	uint32_t dist;
	uint32_t reps[kNumReps];
};

struct Encoder {
	// input
	const uint8_t *data;
	size_t n;
	MatchLists ml;
	size_t mf_pos = 0;  // positions consumed from the finder
	uint64_t mf_off = 0; // running offset of position mf_pos into ml.pairs

	// parameters
	unsigned lc, lp, pb, fast_bytes;
	bool fast_mode = false;
	uint32_t dict_size;
	unsigned pb_mask, dist_table_size;
	uint32_t lp_mask;

	// coder state
	RangeCoder rc;
	unsigned state = 0;
	uint32_t reps[kNumReps];
	unsigned opt_cur = 0, opt_end = 0;
	unsigned longest_len = 0, num_pairs = 0;
	uint32_t num_avail = 0;
	unsigned add_offset = 0;
	uint32_t back_res = 0;
	unsigned match_price_count = 0;
	int rep_len_counter = 0;

	uint32_t prob_prices[kBitModelTotal >> kMoveReducingBits];
	uint32_t matches[kMatchLenMax * 2 + 2];

	uint32_t align_prices[kAlignTableSize];
	uint32_t slot_prices[kNumLenToPosStates][kDistTableSizeMax];
	uint32_t dist_prices[kNumLenToPosStates][kNumFullDistances];

	Prob pos_align[1 << kNumAlignBits];
	Prob is_rep[kNumStates], is_rep_g0[kNumStates], is_rep_g1[kNumStates], is_rep_g2[kNumStates];
	Prob is_match[kNumStates][kPbStatesMax];
	Prob is_rep0_long[kNumStates][kPbStatesMax];
	Prob slot_enc[kNumLenToPosStates][1 << kNumPosSlotBits];
	Prob pos_enc[kNumFullDistances];
	LenProbs len_probs, rep_len_probs;
	LenPrices len_prices, rep_len_prices;
	std::vector<Prob> lit_probs;
	Opt opt[kNumOpts];

	// ---- prices -----------------------------------------------------
	inline uint32_t price(unsigned prob, unsigned bit) const
	{
		return prob_prices[(prob ^ (unsigned)((0 - (int)bit) & (kBitModelTotal - 1))) >> kMoveReducingBits];
	}
	inline uint32_t price0(unsigned prob) const { return prob_prices[prob >> kMoveReducingBits]; }
	inline uint32_t price1(unsigned prob) const { return prob_prices[(prob ^ (kBitModelTotal - 1)) >> kMoveReducingBits]; }

	void init_prob_prices()
	{
		for (uint32_t i = 0; i < (kBitModelTotal >> kMoveReducingBits); i++) {
			uint32_t w = (i << kMoveReducingBits) + (1u << (kMoveReducingBits - 1));
			unsigned bits = 0;
			for (unsigned j = 0; j < kBitPriceShift; j++) {
				w = w * w;
				bits <<= 1;
				while (w >= (1u << 16)) {
					w >>= 1;
					bits++;
				}
			}
			prob_prices[i] = (kBitModelBits << kBitPriceShift) - 15 - bits;
		}
	}

	uint32_t lit_price(const Prob *probs, uint32_t sym) const
	{
		uint32_t pr = 0;
		sym |= 0x100;
		do {
			unsigned b = sym & 1;
			sym >>= 1;
			pr += price(probs[sym], b);
		} while (sym >= 2);
		return pr;
	}
	uint32_t lit_price_matched(const Prob *probs, uint32_t sym, uint32_t match_byte) const
	{
		uint32_t pr = 0, offs = 0x100;
		sym |= 0x100;
		do {
			match_byte <<= 1;
			pr += price(probs[offs + (match_byte & offs) + (sym >> 8)], (sym >> 7) & 1);
			sym <<= 1;
			offs &= ~(match_byte ^ sym);
		} while (sym < 0x10000);
		return pr;
	}
	inline Prob *lit_ctx(uint32_t pos, unsigned prev)
	{
		return lit_probs.data() + (size_t)3 * ((((pos << 8) + prev) & lp_mask) << lc);
	}

	void set_prices3(const Prob *probs, uint32_t start, uint32_t *prices) const
	{
		for (unsigned i = 0; i < 8; i += 2) {
			uint32_t pr = start;
			pr += price(probs[1], i >> 2);
			pr += price(probs[2 + (i >> 2)], (i >> 1) & 1);
			unsigned prob = probs[4 + (i >> 1)];
			prices[i] = pr + price0(prob);
			prices[i + 1] = pr + price1(prob);
		}
	}
	void update_len_prices(LenPrices &lp_, const LenProbs &enc) const
	{
		const unsigned num_pos_states = 1u << pb;
		uint32_t b;
		{
			unsigned prob = enc.low[0];
			b = price1(prob);
			uint32_t a = price0(prob);
			uint32_t c = b + price0(enc.low[kLenLowSymbols]);
			for (unsigned ps = 0; ps < num_pos_states; ps++) {
				uint32_t *prices = lp_.prices[ps];
				const Prob *probs = enc.low + (ps << (1 + kLenLowBits));
				set_prices3(probs, a, prices);
				set_prices3(probs + kLenLowSymbols, c, prices + kLenLowSymbols);
			}
		}
		unsigned i = lp_.table_size;
		if (i > kLenLowSymbols * 2) {
			const Prob *probs = enc.high;
			uint32_t *prices = lp_.prices[0] + kLenLowSymbols * 2;
			i -= kLenLowSymbols * 2 - 1;
			i >>= 1;
			b += price1(enc.low[kLenLowSymbols]);
			do {
				unsigned sym = --i + (1u << (kLenHighBits - 1));
				uint32_t pr = b;
				do {
					unsigned bit = sym & 1;
					sym >>= 1;
					pr += price(probs[sym], bit);
				} while (sym >= 2);
				unsigned prob = probs[(size_t)i + (1u << (kLenHighBits - 1))];
				prices[(size_t)i * 2] = pr + price0(prob);
				prices[(size_t)i * 2 + 1] = pr + price1(prob);
			} while (i);
			size_t num = (lp_.table_size - kLenLowSymbols * 2) * sizeof(uint32_t);
			for (unsigned ps = 1; ps < num_pos_states; ps++)
				memcpy(lp_.prices[ps] + kLenLowSymbols * 2, lp_.prices[0] + kLenLowSymbols * 2, num);
		}
	}
	inline uint32_t len_price(const LenPrices &t, unsigned pos_state, unsigned len) const
	{
		return t.prices[pos_state][len - kMatchLenMin];
	}

	void fill_align_prices()
	{
		for (unsigned i = 0; i < kAlignTableSize / 2; i++) {
			uint32_t pr = 0;
			unsigned sym = i, m = 1, bit;
			bit = sym & 1; sym >>= 1; pr += price(pos_align[m], bit); m = (m << 1) + bit;
			bit = sym & 1; sym >>= 1; pr += price(pos_align[m], bit); m = (m << 1) + bit;
			bit = sym & 1; sym >>= 1; pr += price(pos_align[m], bit); m = (m << 1) + bit;
			unsigned prob = pos_align[m];
			align_prices[i] = pr + price0(prob);
			align_prices[i + 8] = pr + price1(prob);
		}
	}
	void fill_distance_prices()
	{
		uint32_t temp[kNumFullDistances];
		match_price_count = 0;
		for (unsigned i = kStartPosModelIndex / 2; i < kNumFullDistances / 2; i++) {
			unsigned slot = pos_slot(i);
			unsigned footer = (slot >> 1) - 1;
			unsigned base = (2 | (slot & 1)) << footer;
			const Prob *probs = pos_enc + (size_t)base * 2;
			uint32_t pr = 0;
			unsigned m = 1, sym = i, offset = 1u << footer;
			base += i;
			if (footer)
				do {
					unsigned bit = sym & 1;
					sym >>= 1;
					pr += price(probs[m], bit);
					m = (m << 1) + bit;
				} while (--footer);
			unsigned prob = probs[m];
			temp[base] = pr + price0(prob);
			temp[base + offset] = pr + price1(prob);
		}
		for (unsigned lps = 0; lps < kNumLenToPosStates; lps++) {
			unsigned half = (dist_table_size + 1) >> 1;
			uint32_t *sp = slot_prices[lps];
			const Prob *probs = slot_enc[lps];
			for (unsigned slot = 0; slot < half; slot++) {
				unsigned sym = slot + (1u << (kNumPosSlotBits - 1)), bit;
				uint32_t pr;
				bit = sym & 1; sym >>= 1; pr = price(probs[sym], bit);
				bit = sym & 1; sym >>= 1; pr += price(probs[sym], bit);
				bit = sym & 1; sym >>= 1; pr += price(probs[sym], bit);
				bit = sym & 1; sym >>= 1; pr += price(probs[sym], bit);
				bit = sym & 1; sym >>= 1; pr += price(probs[sym], bit);
				unsigned prob = probs[(size_t)slot + (1u << (kNumPosSlotBits - 1))];
				sp[(size_t)slot * 2] = pr + price0(prob);
				sp[(size_t)slot * 2 + 1] = pr + price1(prob);
			}
			uint32_t delta = (uint32_t)((kEndPosModelIndex / 2 - 1) - kNumAlignBits) << kBitPriceShift;
			for (unsigned slot = kEndPosModelIndex / 2; slot < half; slot++) {
				sp[(size_t)slot * 2] += delta;
				sp[(size_t)slot * 2 + 1] += delta;
				delta += 1u << kBitPriceShift;
			}
			uint32_t *dp = dist_prices[lps];
			dp[0] = sp[0];
			dp[1] = sp[1];
			dp[2] = sp[2];
			dp[3] = sp[3];
			for (unsigned i = 4; i < kNumFullDistances; i += 2) {
				uint32_t s = sp[pos_slot(i)];
				dp[i] = s + temp[i];
				dp[i + 1] = s + temp[i + 1];
			}
		}
	}

	// ---- symbol coders ----------------------------------------------
	void enc_literal(Prob *probs, uint32_t sym)
	{
		sym |= 0x100;
		do {
			Prob *pr = probs + (sym >> 8);
			uint32_t b = (sym >> 7) & 1;
			sym <<= 1;
			rc.bit(pr, b);
		} while (sym < 0x10000);
	}
	void enc_literal_matched(Prob *probs, uint32_t sym, uint32_t match_byte)
	{
		uint32_t offs = 0x100;
		sym |= 0x100;
		do {
			match_byte <<= 1;
			Prob *pr = probs + (offs + (match_byte & offs) + (sym >> 8));
			uint32_t b = (sym >> 7) & 1;
			sym <<= 1;
			offs &= ~(match_byte ^ sym);
			rc.bit(pr, b);
		} while (sym < 0x10000);
	}
	void enc_len(LenProbs &lpz, unsigned sym, unsigned pos_state)
	{
		Prob *probs = lpz.low;
		if (sym >= kLenLowSymbols) {
			rc.bit(probs, 1);
			probs += kLenLowSymbols;
			if (sym >= kLenLowSymbols * 2) {
				rc.bit(probs, 1);
				enc_literal(lpz.high, sym - kLenLowSymbols * 2);
				return;
			}
			sym -= kLenLowSymbols;
		}
		rc.bit(probs, 0);
		probs += pos_state << (1 + kLenLowBits);
		unsigned m, b;
		b = sym >> 2;       rc.bit(probs + 1, b); m = (1 << 1) + b;
		b = (sym >> 1) & 1; rc.bit(probs + m, b); m = (m << 1) + b;
		b = sym & 1;        rc.bit(probs + m, b);
	}
	void enc_reverse(Prob *probs, unsigned nbits, unsigned sym)
	{
		unsigned m = 1;
		do {
			unsigned b = sym & 1;
			sym >>= 1;
			rc.bit(probs + m, b);
			m = (m << 1) | b;
		} while (--nbits);
	}

	// ---- match finder plumbing --------------------------------------
	inline const uint8_t *cur_ptr() const { return data + mf_pos; }
	inline uint32_t avail_now() const { return (uint32_t)(n - mf_pos); }
	inline void move_pos(unsigned num)
	{
		add_offset += num;
		if (ml.wait_ready)
			ml.wait_ready(ml.ctx, mf_pos + num - 1);
		for (unsigned k = 0; k < num; k++)
			mf_off += ml.counts[mf_pos + k];
		mf_pos += num;
	}
	unsigned read_matches(unsigned *num_pairs_res)
	{
		add_offset++;
		num_avail = avail_now();
		if (ml.wait_ready)
			ml.wait_ready(ml.ctx, mf_pos);
		unsigned np = ml.counts[mf_pos];
		if (ml.packed) {
			const uint32_t *src = ml.pairs + (mf_off >> 1);
			for (unsigned k = 0; k < np; k += 2) {
				const uint32_t v = src[k >> 1];
				matches[k] = v >> 25;
				matches[k + 1] = v & 0x1FFFFFFu;
			}
		} else
			memcpy(matches, ml.pairs + mf_off, (size_t)np * 4);
		mf_off += np;
		mf_pos++;
		*num_pairs_res = np;
		if (np == 0)
			return 0;
		unsigned len = matches[np - 2];
		if (len != fast_bytes)
			return len;
		uint32_t na = num_avail > kMatchLenMax ? kMatchLenMax : num_avail;
		const uint8_t *p1 = cur_ptr() - 1;
		const uint8_t *p2 = p1 + len;
		ptrdiff_t dif = (ptrdiff_t)-1 - (ptrdiff_t)matches[np - 1];
		const uint8_t *lim = p1 + na;
		for (; p2 != lim && *p2 == p2[dif]; p2++) {
		}
		return (unsigned)(p2 - p1);
	}

	// ---- optimal parser ---------------------------------------------
	inline uint32_t price_short_rep(unsigned st, unsigned ps) const
	{
		return price0(is_rep_g0[st]) + price0(is_rep0_long[st][ps]);
	}
	inline uint32_t price_rep0(unsigned st, unsigned ps) const
	{
		return price1(is_match[st][ps]) + price1(is_rep0_long[st][ps]) + price1(is_rep[st]) + price0(is_rep_g0[st]);
	}
	inline uint32_t price_pure_rep(unsigned rep_index, unsigned st, unsigned ps) const
	{
		uint32_t pr;
		unsigned prob = is_rep_g0[st];
		if (rep_index == 0) {
			pr = price0(prob);
			pr += price1(is_rep0_long[st][ps]);
		} else {
			pr = price1(prob);
			prob = is_rep_g1[st];
			if (rep_index == 1)
				pr += price0(prob);
			else {
				pr += price1(prob);
				pr += price(is_rep_g2[st], rep_index - 2);
			}
		}
		return pr;
	}

	unsigned backward(unsigned cur)
	{
		unsigned wr = cur + 1;
		opt_end = wr;
		for (;;) {
			uint32_t dist = opt[cur].dist;
			unsigned len = opt[cur].len;
			unsigned extra = opt[cur].extra;
			cur -= len;
			if (extra) {
				wr--;
				opt[wr].len = len;
				cur -= extra;
				len = extra;
				if (extra == 1) {
					opt[wr].dist = dist;
					dist = kMarkLit;
				} else {
					opt[wr].dist = 0;
					len--;
					wr--;
					opt[wr].dist = kMarkLit;
					opt[wr].len = 1;
				}
			}
			if (cur == 0) {
				back_res = dist;
				opt_cur = wr;
				return len;
			}
			wr--;
			opt[wr].dist = dist;
			opt[wr].len = len;
		}
	}

	unsigned optimum(uint32_t position)
	{
		unsigned last, cur;
		uint32_t rp[kNumReps];
		unsigned rep_lens[kNumReps];
		uint32_t *mt = matches;

		{
			uint32_t navail;
			unsigned npairs, main_len, rep_max = 0, i, pos_state;
			uint32_t match_price, rep_match_price;
			const uint8_t *d;
			uint8_t cur_byte, match_byte;

			opt_cur = opt_end = 0;
			if (add_offset == 0)
				main_len = read_matches(&npairs);
			else {
				main_len = longest_len;
				npairs = num_pairs;
			}
			navail = num_avail;
			if (navail < 2) {
				back_res = kMarkLit;
				return 1;
			}
			if (navail > kMatchLenMax)
				navail = kMatchLenMax;

			d = cur_ptr() - 1;
			for (i = 0; i < kNumReps; i++) {
				rp[i] = reps[i];
				const uint8_t *d2 = d - rp[i];
				if (d[0] != d2[0] || d[1] != d2[1]) {
					rep_lens[i] = 0;
					continue;
				}
				unsigned len;
				len = mismatch_from(d, d2, 2, navail);
				rep_lens[i] = len;
				if (len > rep_lens[rep_max])
					rep_max = i;
				if (len == kMatchLenMax)
					break;
			}
			// the reference leaves rep_lens[j] unset for j after an early break; it only reads
			// rep_lens[rep_max] and rep_lens[0..] below when no early exit happened.
			if (i < kNumReps)
				for (unsigned j = i + 1; j < kNumReps; j++)
					rep_lens[j] = 0;

			if (rep_lens[rep_max] >= fast_bytes) {
				back_res = rep_max;
				unsigned len = rep_lens[rep_max];
				move_pos(len - 1);
				return len;
			}
			if (main_len >= fast_bytes) {
				back_res = mt[(size_t)npairs - 1] + kNumReps;
				move_pos(main_len - 1);
				return main_len;
			}

			cur_byte = *d;
			match_byte = *(d - rp[0]);
			last = rep_lens[rep_max];
			if (last <= main_len)
				last = main_len;
			if (last < 2 && cur_byte != match_byte) {
				back_res = kMarkLit;
				return 1;
			}

			opt[0].state = (uint16_t)state;
			pos_state = position & pb_mask;
			{
				const Prob *probs = lit_ctx(position, *(d - 1));
				opt[1].price = price0(is_match[state][pos_state]) +
					       (!is_lit_state(state) ? lit_price_matched(probs, cur_byte, match_byte)
								     : lit_price(probs, cur_byte));
			}
			opt[1].dist = kMarkLit;
			opt[1].extra = 0;

			match_price = price1(is_match[state][pos_state]);
			rep_match_price = match_price + price1(is_rep[state]);

			if (match_byte == cur_byte && rep_lens[0] == 0) {
				uint32_t sp = rep_match_price + price_short_rep(state, pos_state);
				if (sp < opt[1].price) {
					opt[1].price = sp;
					opt[1].dist = 0;
					opt[1].extra = 0;
				}
				if (last < 2) {
					back_res = opt[1].dist;
					return 1;
				}
			}
			opt[1].len = 1;
			opt[0].reps[0] = rp[0];
			opt[0].reps[1] = rp[1];
			opt[0].reps[2] = rp[2];
			opt[0].reps[3] = rp[3];

			for (i = 0; i < kNumReps; i++) {
				unsigned rl = rep_lens[i];
				if (rl < 2)
					continue;
				uint32_t pr = rep_match_price + price_pure_rep(i, state, pos_state);
				do {
					uint32_t p2 = pr + len_price(rep_len_prices, pos_state, rl);
					Opt *o = &opt[rl];
					if (p2 < o->price) {
						o->price = p2;
						o->len = rl;
						o->dist = i;
						o->extra = 0;
					}
				} while (--rl >= 2);
			}

			{
				unsigned len = rep_lens[0] + 1;
				if (len <= main_len) {
					unsigned offs = 0;
					uint32_t normal = match_price + price0(is_rep[state]);
					if (len < 2)
						len = 2;
					else
						while (len > mt[offs])
							offs += 2;
					for (;; len++) {
						uint32_t dist = mt[(size_t)offs + 1];
						uint32_t pr = normal + len_price(len_prices, pos_state, len);
						unsigned lps = len_to_pos_state(len);
						if (dist < kNumFullDistances)
							pr += dist_prices[lps][dist & (kNumFullDistances - 1)];
						else {
							unsigned slot = pos_slot(dist);
							pr += align_prices[dist & kAlignMask];
							pr += slot_prices[lps][slot];
						}
						Opt *o = &opt[len];
						if (pr < o->price) {
							o->price = pr;
							o->len = len;
							o->dist = dist + kNumReps;
							o->extra = 0;
						}
						if (len == mt[offs]) {
							offs += 2;
							if (offs == npairs)
								break;
						}
					}
				}
			}
			cur = 0;
		}

		for (;;) {
			unsigned navail;
			uint32_t navail_full;
			unsigned new_len, npairs, prev, st, pos_state, start_len;
			uint32_t lit_pr, match_price, rep_match_price;
			bool next_is_lit;
			uint8_t cur_byte, match_byte;
			const uint8_t *d;
			Opt *co, *no;

			if (++cur == last)
				break;

			if (cur >= kNumOpts - 64) {
				unsigned best = cur;
				uint32_t pr = opt[cur].price;
				for (unsigned j = cur + 1; j <= last; j++) {
					uint32_t p2 = opt[j].price;
					if (pr >= p2) {
						pr = p2;
						best = j;
					}
				}
				unsigned delta = best - cur;
				if (delta != 0)
					move_pos(delta);
				cur = best;
				break;
			}

			new_len = read_matches(&npairs);
			if (new_len >= fast_bytes) {
				num_pairs = npairs;
				longest_len = new_len;
				break;
			}

			co = &opt[cur];
			position++;
			prev = cur - co->len;

			if (co->len == 1) {
				st = opt[prev].state;
				st = co->dist == 0 ? kShortRepNext[st] : kLitNext[st];
			} else {
				uint32_t dist = co->dist;
				if (co->extra) {
					prev -= co->extra;
					st = kStateRepAfterLit;
					if (co->extra == 1)
						st = dist < kNumReps ? kStateRepAfterLit : kStateMatchAfterLit;
				} else {
					st = opt[prev].state;
					st = dist < kNumReps ? kRepNext[st] : kMatchNext[st];
				}
				const Opt *po = &opt[prev];
				uint32_t b0 = po->reps[0];
				if (dist < kNumReps) {
					if (dist == 0) {
						rp[0] = b0;
						rp[1] = po->reps[1];
						rp[2] = po->reps[2];
						rp[3] = po->reps[3];
					} else {
						rp[1] = b0;
						b0 = po->reps[1];
						if (dist == 1) {
							rp[0] = b0;
							rp[2] = po->reps[2];
							rp[3] = po->reps[3];
						} else {
							rp[2] = b0;
							rp[0] = po->reps[dist];
							rp[3] = po->reps[dist ^ 1];
						}
					}
				} else {
					rp[0] = dist - kNumReps + 1;
					rp[1] = b0;
					rp[2] = po->reps[1];
					rp[3] = po->reps[2];
				}
			}

			co->state = (uint16_t)st;
			co->reps[0] = rp[0];
			co->reps[1] = rp[1];
			co->reps[2] = rp[2];
			co->reps[3] = rp[3];

			d = cur_ptr() - 1;
			cur_byte = *d;
			match_byte = *(d - rp[0]);
			pos_state = position & pb_mask;

			{
				uint32_t cp = co->price;
				unsigned prob = is_match[st][pos_state];
				match_price = cp + price1(prob);
				lit_pr = cp + price0(prob);
			}

			no = &opt[(size_t)cur + 1];
			next_is_lit = false;

			if ((no->price < kInfinityPrice && match_byte == cur_byte) || lit_pr > no->price)
				lit_pr = 0;
			else {
				const Prob *probs = lit_ctx(position, *(d - 1));
				lit_pr += !is_lit_state(st) ? lit_price_matched(probs, cur_byte, match_byte) : lit_price(probs, cur_byte);
				if (lit_pr < no->price) {
					no->price = lit_pr;
					no->len = 1;
					no->dist = kMarkLit;
					no->extra = 0;
					next_is_lit = true;
				}
			}

			rep_match_price = match_price + price1(is_rep[st]);

			navail_full = num_avail;
			{
				unsigned temp = kNumOpts - 1 - cur;
				if (navail_full > temp)
					navail_full = temp;
			}

			if (is_lit_state(st) && match_byte == cur_byte && rep_match_price < no->price &&
			    (no->len < 2 || no->dist != 0)) {
				uint32_t sp = rep_match_price + price_short_rep(st, pos_state);
				if (sp < no->price) {
					no->price = sp;
					no->len = 1;
					no->dist = 0;
					no->extra = 0;
					next_is_lit = false;
				}
			}

			if (navail_full < 2)
				continue;
			navail = navail_full <= fast_bytes ? navail_full : fast_bytes;

			// LIT : REP_0
			if (!next_is_lit && lit_pr != 0 && match_byte != cur_byte && navail_full > 2) {
				const uint8_t *d2 = d - rp[0];
				if (d[1] == d2[1] && d[2] == d2[2]) {
					unsigned len, limit = fast_bytes + 1;
					if (limit > navail_full)
						limit = navail_full;
					len = mismatch_from(d, d2, 3, limit);
					unsigned st2 = kLitNext[st];
					unsigned ps2 = (position + 1) & pb_mask;
					uint32_t pr = lit_pr + price_rep0(st2, ps2);
					unsigned offset = cur + len;
					if (last < offset)
						last = offset;
					len--;
					uint32_t p2 = pr + len_price(rep_len_prices, ps2, len);
					Opt *o = &opt[offset];
					if (p2 < o->price) {
						o->price = p2;
						o->len = len;
						o->dist = 0;
						o->extra = 1;
					}
				}
			}

			start_len = 2;

			// REP
			for (unsigned ri = 0; ri < kNumReps; ri++) {
				const uint8_t *d2 = d - rp[ri];
				if (d[0] != d2[0] || d[1] != d2[1])
					continue;
				unsigned len;
				len = mismatch_from(d, d2, 2, navail);
				{
					unsigned offset = cur + len;
					if (last < offset)
						last = offset;
				}
				uint32_t pr;
				{
					unsigned l2 = len;
					pr = rep_match_price + price_pure_rep(ri, st, pos_state);
					do {
						uint32_t p2 = pr + len_price(rep_len_prices, pos_state, l2);
						Opt *o = &opt[cur + l2];
						if (p2 < o->price) {
							o->price = p2;
							o->len = l2;
							o->dist = ri;
							o->extra = 0;
						}
					} while (--l2 >= 2);
				}
				if (ri == 0)
					start_len = len + 1;

				// REP : LIT : REP_0
				{
					unsigned l2 = len + 1;
					unsigned limit = l2 + fast_bytes;
					if (limit > navail_full)
						limit = navail_full;
					l2 += 2;
					if (l2 <= limit && d[l2 - 2] == d2[l2 - 2] && d[l2 - 1] == d2[l2 - 1]) {
						unsigned st2 = kRepNext[st];
						unsigned ps2 = (position + len) & pb_mask;
						pr += len_price(rep_len_prices, pos_state, len) + price0(is_match[st2][ps2]) +
						      lit_price_matched(lit_ctx(position + len, d[(size_t)len - 1]), d[len], d2[len]);
						st2 = kStateLitAfterRep;
						ps2 = (ps2 + 1) & pb_mask;
						pr += price_rep0(st2, ps2);
						l2 = mismatch_from(d, d2, l2, limit);
						l2 -= len;
						unsigned offset = cur + len + l2;
						if (last < offset)
							last = offset;
						l2--;
						uint32_t p2 = pr + len_price(rep_len_prices, ps2, l2);
						Opt *o = &opt[offset];
						if (p2 < o->price) {
							o->price = p2;
							o->len = l2;
							o->extra = (uint16_t)(len + 1);
							o->dist = ri;
						}
					}
				}
			}

			// MATCH
			if (new_len > navail) {
				new_len = navail;
				for (npairs = 0; new_len > mt[npairs]; npairs += 2) {
				}
				mt[npairs] = new_len;
				npairs += 2;
			}

			if (new_len >= start_len) {
				uint32_t normal = match_price + price0(is_rep[st]);
				uint32_t dist;
				unsigned offs, slot, len;
				{
					unsigned offset = cur + new_len;
					if (last < offset)
						last = offset;
				}
				offs = 0;
				while (start_len > mt[offs])
					offs += 2;
				dist = mt[(size_t)offs + 1];
				slot = pos_slot(dist);

				for (len = start_len;; len++) {
					uint32_t pr = normal + len_price(len_prices, pos_state, len);
					{
						unsigned ln = len - 2;
						ln = ln < kNumLenToPosStates - 1 ? ln : kNumLenToPosStates - 1;
						if (dist < kNumFullDistances)
							pr += dist_prices[ln][dist & (kNumFullDistances - 1)];
						else
							pr += slot_prices[ln][slot] + align_prices[dist & kAlignMask];
						Opt *o = &opt[cur + len];
						if (pr < o->price) {
							o->price = pr;
							o->len = len;
							o->dist = dist + kNumReps;
							o->extra = 0;
						}
					}
					if (len == mt[offs]) {
						// MATCH : LIT : REP_0
						const uint8_t *d2 = d - dist - 1;
						unsigned l2 = len + 1;
						unsigned limit = l2 + fast_bytes;
						if (limit > navail_full)
							limit = navail_full;
						l2 += 2;
						if (l2 <= limit && d[l2 - 2] == d2[l2 - 2] && d[l2 - 1] == d2[l2 - 1]) {
							l2 = mismatch_from(d, d2, l2, limit);
							l2 -= len;
							unsigned st2 = kMatchNext[st];
							unsigned ps2 = (position + len) & pb_mask;
							pr += price0(is_match[st2][ps2]);
							pr += lit_price_matched(lit_ctx(position + len, d[(size_t)len - 1]), d[len], d2[len]);
							st2 = kStateLitAfterMatch;
							ps2 = (ps2 + 1) & pb_mask;
							pr += price_rep0(st2, ps2);
							unsigned offset = cur + len + l2;
							if (last < offset)
								last = offset;
							l2--;
							uint32_t p2 = pr + len_price(rep_len_prices, ps2, l2);
							Opt *o = &opt[offset];
							if (p2 < o->price) {
								o->price = p2;
								o->len = l2;
								o->extra = (uint16_t)(len + 1);
								o->dist = dist + kNumReps;
							}
						}
						offs += 2;
						if (offs == npairs)
							break;
						dist = mt[(size_t)offs + 1];
						slot = pos_slot(dist);
					}
				}
			}
		}

		do
			opt[last].price = kInfinityPrice;
		while (--last);

		return backward(cur);
	}

	// ---- fast parser (algo 0, levels 1-4): GetOptimumFast, LzmaEnc.c:1970-2098 -------------------
	// Greedy with one position of look-ahead: take a repeat match if it is nearly as long as the main
	// match (nearer is cheaper), shorten the main match while the next shorter one is >128x nearer, and
	// emit a literal instead when the next position offers something clearly better.
	static inline bool change_pair(uint32_t small_dist, uint32_t big_dist) { return (big_dist >> 7) > small_dist; }
	unsigned optimum_fast()
	{
		unsigned main_len, npairs;
		if (add_offset == 0)
			main_len = read_matches(&npairs);
		else {
			main_len = longest_len;
			npairs = num_pairs;
		}
		uint32_t navail = num_avail;
		back_res = kMarkLit;
		if (navail < 2)
			return 1;
		if (navail > kMatchLenMax)
			navail = kMatchLenMax;
		const uint8_t *d = cur_ptr() - 1;
		unsigned rep_len = 0, rep_index = 0;
		for (unsigned i = 0; i < kNumReps; i++) {
			const uint8_t *d2 = d - reps[i];
			if (d[0] != d2[0] || d[1] != d2[1])
				continue;
			unsigned len;
			len = mismatch_from(d, d2, 2, navail);
			if (len >= fast_bytes) {
				back_res = i;
				move_pos(len - 1);
				return len;
			}
			if (len > rep_len) {
				rep_index = i;
				rep_len = len;
			}
		}
		if (main_len >= fast_bytes) {
			back_res = matches[npairs - 1] + kNumReps;
			move_pos(main_len - 1);
			return main_len;
		}
		uint32_t main_dist = 0;
		if (main_len >= 2) {
			main_dist = matches[npairs - 1];
			while (npairs > 2) {
				if (main_len != matches[npairs - 4] + 1)
					break;
				const uint32_t dist2 = matches[npairs - 3];
				if (!change_pair(dist2, main_dist))
					break;
				npairs -= 2;
				main_len--;
				main_dist = dist2;
			}
			if (main_len == 2 && main_dist >= 0x80)
				main_len = 1;
		}
		if (rep_len >= 2)
			if (rep_len + 1 >= main_len || (rep_len + 2 >= main_len && main_dist >= (1u << 9)) ||
			    (rep_len + 3 >= main_len && main_dist >= (1u << 15))) {
				back_res = rep_index;
				move_pos(rep_len - 1);
				return rep_len;
			}
		if (main_len < 2 || navail <= 2)
			return 1;
		{
			const unsigned len1 = read_matches(&num_pairs);
			longest_len = len1;
			if (len1 >= 2) {
				const uint32_t new_dist = matches[num_pairs - 1];
				if ((len1 >= main_len && new_dist < main_dist) || (len1 == main_len + 1 && !change_pair(main_dist, new_dist)) ||
				    (len1 > main_len + 1) || (len1 + 1 >= main_len && main_len >= 3 && change_pair(new_dist, main_dist)))
					return 1;
			}
		}
		d = cur_ptr() - 1;
		for (unsigned i = 0; i < kNumReps; i++) {
			const uint8_t *d2 = d - reps[i];
			if (d[0] != d2[0] || d[1] != d2[1])
				continue;
			const unsigned limit = main_len - 1;
			for (unsigned len = 2;; len++) {
				if (len >= limit)
					return 1;
				if (d[len] != d2[len])
					break;
			}
		}
		back_res = main_dist + kNumReps;
		if (main_len != 2)
			move_pos(main_len - 2);
		return main_len;
	}

	// ---- init / main loop -------------------------------------------
	void init(const LzmaParams &prm)
	{
		lc = (unsigned)prm.lc;
		lp = (unsigned)prm.lp;
		pb = (unsigned)prm.pb;
		unsigned fb = (unsigned)prm.fb;
		if (fb < 5) fb = 5;
		if (fb > kMatchLenMax) fb = kMatchLenMax;
		fast_bytes = fb;
		fast_mode = prm.fast;
		dict_size = prm.dict_size;
		unsigned i;
		for (i = kEndPosModelIndex / 2; i < 32; i++)
			if (dict_size <= (1u << i))
				break;
		dist_table_size = i * 2;

		state = 0;
		reps[0] = reps[1] = reps[2] = reps[3] = 1;
		for (auto &p : pos_align) p = kProbInit;
		for (unsigned s = 0; s < kNumStates; s++) {
			for (unsigned j = 0; j < kPbStatesMax; j++) {
				is_match[s][j] = kProbInit;
				is_rep0_long[s][j] = kProbInit;
			}
			is_rep[s] = is_rep_g0[s] = is_rep_g1[s] = is_rep_g2[s] = kProbInit;
		}
		for (unsigned s = 0; s < kNumLenToPosStates; s++)
			for (unsigned j = 0; j < (1u << kNumPosSlotBits); j++)
				slot_enc[s][j] = kProbInit;
		for (auto &p : pos_enc) p = kProbInit;
		lit_probs.assign((size_t)0x300 << (lp + lc), kProbInit);
		len_probs.init();
		rep_len_probs.init();
		opt_end = opt_cur = 0;
		for (unsigned k = 0; k < kNumOpts; k++)
			opt[k].price = kInfinityPrice;
		add_offset = 0;
		pb_mask = (1u << pb) - 1;
		lp_mask = (0x100u << lp) - (0x100u >> lc);

		init_prob_prices();
		fill_distance_prices();
		fill_align_prices();
		len_prices.table_size = rep_len_prices.table_size = fast_bytes + 1 - kMatchLenMin;
		rep_len_counter = kRepLenCount;
		update_len_prices(len_prices, len_probs);
		update_len_prices(rep_len_prices, rep_len_probs);
	}

	void run()
	{
		uint32_t now_pos = 0;
		if (n == 0) {
			rc.flush();
			return;
		}
		{
			unsigned np;
			read_matches(&np);
			rc.bit(&is_match[0][0], 0);
			uint8_t cb = *(cur_ptr() - add_offset);
			enc_literal(lit_probs.data(), cb);
			add_offset--;
			now_pos++;
		}
		if (avail_now() != 0)
			for (;;) {
				unsigned len;
				if (fast_mode)
					len = optimum_fast();
				else if (opt_end == opt_cur)
					len = optimum(now_pos);
				else {
					const Opt *o = &opt[opt_cur];
					len = o->len;
					back_res = o->dist;
					opt_cur++;
				}
				unsigned pos_state = now_pos & pb_mask;
				uint32_t dist = back_res;

				if (dist == kMarkLit) {
					rc.bit(&is_match[state][pos_state], 0);
					const uint8_t *dp = cur_ptr() - add_offset;
					Prob *probs = lit_ctx(now_pos, *(dp - 1));
					uint8_t cb = *dp;
					unsigned st = state;
					state = kLitNext[st];
					if (is_lit_state(st))
						enc_literal(probs, cb);
					else
						enc_literal_matched(probs, cb, *(dp - reps[0]));
				} else {
					rc.bit(&is_match[state][pos_state], 1);
					if (dist < kNumReps) {
						rc.bit(&is_rep[state], 1);
						if (dist == 0) {
							rc.bit(&is_rep_g0[state], 0);
							if (len != 1)
								rc.bit(&is_rep0_long[state][pos_state], 1);
							else {
								rc.bit(&is_rep0_long[state][pos_state], 0);
								state = kShortRepNext[state];
							}
						} else {
							rc.bit(&is_rep_g0[state], 1);
							if (dist == 1) {
								rc.bit(&is_rep_g1[state], 0);
								dist = reps[1];
							} else {
								rc.bit(&is_rep_g1[state], 1);
								if (dist == 2) {
									rc.bit(&is_rep_g2[state], 0);
									dist = reps[2];
								} else {
									rc.bit(&is_rep_g2[state], 1);
									dist = reps[3];
									reps[3] = reps[2];
								}
								reps[2] = reps[1];
							}
							reps[1] = reps[0];
							reps[0] = dist;
						}
						if (len != 1) {
							enc_len(rep_len_probs, len - kMatchLenMin, pos_state);
							--rep_len_counter;
							state = kRepNext[state];
						}
					} else {
						rc.bit(&is_rep[state], 0);
						state = kMatchNext[state];
						enc_len(len_probs, len - kMatchLenMin, pos_state);
						dist -= kNumReps;
						reps[3] = reps[2];
						reps[2] = reps[1];
						reps[1] = reps[0];
						reps[0] = dist + 1;
						match_price_count++;
						unsigned slot = pos_slot(dist);
						{
							uint32_t sym = slot + (1u << kNumPosSlotBits);
							Prob *probs = slot_enc[len_to_pos_state(len)];
							do {
								Prob *pr = probs + (sym >> kNumPosSlotBits);
								uint32_t b = (sym >> (kNumPosSlotBits - 1)) & 1;
								sym <<= 1;
								rc.bit(pr, b);
							} while (sym < (1u << (kNumPosSlotBits * 2)));
						}
						if (dist >= kStartPosModelIndex) {
							unsigned footer = (slot >> 1) - 1;
							if (dist < kNumFullDistances) {
								unsigned base = (2 | (slot & 1)) << footer;
								enc_reverse(pos_enc + base, footer, dist);
							} else {
								// high (footer-4) bits of the footer, MSB first, then 4 align bits reversed
								uint32_t red = dist - ((2u | (slot & 1)) << footer);
								rc.direct(red >> kNumAlignBits, footer - kNumAlignBits);
								unsigned m = 1, b;
								b = dist & 1; dist >>= 1; rc.bit(pos_align + m, b); m = (m << 1) + b;
								b = dist & 1; dist >>= 1; rc.bit(pos_align + m, b); m = (m << 1) + b;
								b = dist & 1; dist >>= 1; rc.bit(pos_align + m, b); m = (m << 1) + b;
								b = dist & 1; rc.bit(pos_align + m, b);
							}
						}
					}
				}

				now_pos += len;
				add_offset -= len;

				if (add_offset == 0) {
					if (!fast_mode && match_price_count >= 64) {
						fill_align_prices();
						fill_distance_prices();
						update_len_prices(len_prices, len_probs);
					}
					if (!fast_mode && rep_len_counter <= 0) {
						rep_len_counter = kRepLenCount;
						update_len_prices(rep_len_prices, rep_len_probs);
					}
					if (avail_now() == 0)
						break;
					if (rc.overflow)
						break; // result is LZ_ERROR_OUTPUT_EOF whatever follows
				}
			}
		rc.flush();
	}
};

} // namespace

uint32_t lzma_hash_mask(uint32_t dict_size, uint64_t expected_size)
{
	uint32_t res[2];
	uint64_t in[2] = {dict_size, expected_size < dict_size ? expected_size : dict_size};
	for (int k = 0; k < 2; k++) {
		uint32_t hs = (uint32_t)in[k];
		if (hs != 0)
			hs--;
		hs |= hs >> 1;
		hs |= hs >> 2;
		hs |= hs >> 4;
		hs |= hs >> 8;
		hs >>= 1;
		if (hs >= (1u << 24))
			hs >>= 1; // numHashBytes == 4
		hs |= 0xFFFF;
		res[k] = hs;
	}
	return res[1] > res[0] ? res[0] : res[1];
}

// same for the 5-byte hash of the HC5 finder (levels 1-4): the low 18 bits are always set
uint32_t lzma_hash_mask5(uint32_t dict_size, uint64_t expected_size)
{
	uint32_t res[2];
	uint64_t in[2] = {dict_size, expected_size < dict_size ? expected_size : dict_size};
	for (int k = 0; k < 2; k++) {
		uint32_t hs = (uint32_t)in[k];
		if (hs != 0)
			hs--;
		hs |= hs >> 1;
		hs |= hs >> 2;
		hs |= hs >> 4;
		hs |= hs >> 8;
		hs >>= 1;
		if (hs >= (1u << 24))
			hs >>= 1;
		hs |= 0xFFFF;
		hs |= (256u << 10) - 1; // kLzHash_CrcShift_2
		res[k] = hs;
	}
	return res[1] > res[0] ? res[0] : res[1];
}

void lzma_write_props(const LzmaParams &prm, uint8_t props[5])
{
	uint32_t dict = prm.dict_size, v;
	props[0] = (uint8_t)((prm.pb * 5 + prm.lp) * 9 + prm.lc);
	if (dict >= (1u << 21)) {
		const uint32_t mask = (1u << 20) - 1;
		v = (dict + mask) & ~mask;
		if (v < dict)
			v = dict;
	} else {
		unsigned i = 11 * 2;
		do {
			v = (uint32_t)(2 + (i & 1)) << (i >> 1);
			i++;
		} while (v < dict);
	}
	props[1] = (uint8_t)v;
	props[2] = (uint8_t)(v >> 8);
	props[3] = (uint8_t)(v >> 16);
	props[4] = (uint8_t)(v >> 24);
}

int lzma_encode_block(const LzmaParams &prm, const uint8_t *src, size_t n, const MatchLists &ml,
		      uint8_t *dest, size_t dest_cap, size_t *dest_len)
{
	if (prm.lc > 8 || prm.lp > 4 || prm.pb > 4 || prm.lc < 0 || prm.lp < 0 || prm.pb < 0)
		return LZ_ERROR_PARAM;
	if ((prm.level < 5) != prm.fast) // algo 0 <=> levels 1-4 here (LzmaEncProps_Normalize)
		return LZ_ERROR_PARAM;
	if (n >= 0xFFFFFFFFu)
		return LZ_ERROR_PARAM;
	std::unique_ptr<Encoder> e(new (std::nothrow) Encoder());
	if (!e)
		return LZ_ERROR_MEM;
	e->data = src;
	e->n = n;
	e->ml = ml;
	e->rc.out = dest;
	e->rc.cap = dest_cap;
	e->init(prm);
	e->run();
	if (e->rc.overflow) {
		*dest_len = dest_cap;
		return LZ_ERROR_OUTPUT_EOF;
	}
	*dest_len = e->rc.len;
	return LZ_OK;
}

} // namespace lrzgpu
