// lzma_model.h -- the adaptive probability model of an LZMA stream and the price (cost) tables the
// optimal parser reads from it.
//
// The model is the format's (the decoder in lzma_dec.cpp keeps the same contexts); reference for the
// numbers a price must come out at: src/lzma/C/LzmaEnc.c:830-897 (bit prices), 963-1065 (length
// prices), 2202-2320 (distance / align prices).  A price is -log2(probability) in 1/16 bit units,
// looked up from the 11-bit probability reduced to 7 bits.  Tables are laid out for how the parser
// walks them: length prices are indexed by the LENGTH itself and contiguous in it, so that "every
// length from a to b of one candidate" is one run of adjacent words.
#pragma once
#include <cstdint>
#include <cstring>
#include <vector>

#include "lzma_rangecoder.h"

namespace lrzgpu {

constexpr unsigned kStates = 12;
constexpr unsigned kPosStatesMax = 16;
constexpr unsigned kLenMin = 2, kLenMax = 273;
constexpr unsigned kLenToDistStates = 4;
constexpr unsigned kDistSlots = 64;
constexpr unsigned kNearDistances = 128; // distances below this are priced through one table
constexpr unsigned kAlignBits = 4, kAlignSize = 1u << kAlignBits;
constexpr unsigned kPriceReduce = 4;     // probability bits dropped before the price lookup
constexpr unsigned kPriceBitShift = 4;   // prices are in 1/16 bit
constexpr uint32_t kPriceInfinite = 1u << 30;

// state machine of the format: what the last few symbols were (LZMA spec; LzmaEnc.c:620-623)
inline unsigned after_literal(unsigned s) { return s < 4 ? 0 : s < 10 ? s - 3 : s - 6; }
inline unsigned after_match(unsigned s) { return s < 7 ? 7 : 10; }
inline unsigned after_rep(unsigned s) { return s < 7 ? 8 : 11; }
inline unsigned after_short_rep(unsigned s) { return s < 7 ? 9 : 11; }
inline bool last_was_literal(unsigned s) { return s < 7; }

inline unsigned dist_slot(uint32_t d) // 2 * floor(log2 d) + the bit below the top one
{
	if (d < 2)
		return d;
	const unsigned i = 31 - (unsigned)__builtin_clz(d);
	return (i << 1) + ((d >> (i - 1)) & 1);
}
inline unsigned len_dist_state(unsigned len) { return len < kLenToDistStates + kLenMin ? len - kLenMin : kLenToDistStates - 1; }

struct BitPrices {
	uint32_t t[kProbOne >> kPriceReduce];
	BitPrices()
	{
		// price(p) = round(-16 log2 p): square the midpoint of the bucket four times, counting the
		// doublings it takes to stay below 2^16 (LzmaEnc.c:830-856)
		for (uint32_t i = 0; i < (kProbOne >> kPriceReduce); i++) {
			uint32_t w = (i << kPriceReduce) + (1u << (kPriceReduce - 1));
			unsigned bits = 0;
			for (unsigned j = 0; j < kPriceBitShift; j++) {
				w *= w;
				bits <<= 1;
				while (w >= (1u << 16)) {
					w >>= 1;
					bits++;
				}
			}
			t[i] = (kProbBits << kPriceBitShift) - 15 - bits;
		}
	}
	inline uint32_t zero(unsigned p) const { return t[p >> kPriceReduce]; }
	inline uint32_t one(unsigned p) const { return t[(p ^ (kProbOne - 1)) >> kPriceReduce]; }
	inline uint32_t bit(unsigned p, unsigned b) const { return t[(p ^ ((0u - b) & (kProbOne - 1))) >> kPriceReduce]; }
};

// length coder: choice bits, two 3-bit trees per position state, one shared 8-bit tree
struct LenModel {
	Prob choice, choice2;
	Prob low[kPosStatesMax][8], mid[kPosStatesMax][8], high[256];
	void reset()
	{
		choice = choice2 = kProbHalf;
		for (auto &r : low)
			for (Prob &p : r)
				p = kProbHalf;
		for (auto &r : mid)
			for (Prob &p : r)
				p = kProbHalf;
		for (Prob &p : high)
			p = kProbHalf;
	}
	void encode(RangeEncoder &rc, unsigned len, unsigned pos_state)
	{
		unsigned s = len - kLenMin;
		if (s < 8) {
			rc.encode(&choice, 0);
			rc.encode_tree<3>(low[pos_state], s);
		} else if (s < 16) {
			rc.encode(&choice, 1);
			rc.encode(&choice2, 0);
			rc.encode_tree<3>(mid[pos_state], s - 8);
		} else {
			rc.encode(&choice, 1);
			rc.encode(&choice2, 1);
			rc.encode_tree<8>(high, s - 16);
		}
	}
};

// prices of every length 2..max_len per position state; row[len] is the price of `len`
struct LenPrices {
	uint32_t row[kPosStatesMax][kLenMax + 1 + 7]; // + slack: the parser loads 8 adjacent words
	void refresh(const LenModel &m, const BitPrices &bp, unsigned pos_states, unsigned max_len)
	{
		const uint32_t c0 = bp.zero(m.choice), c1 = bp.one(m.choice);
		const uint32_t c10 = c1 + bp.zero(m.choice2), c11 = c1 + bp.one(m.choice2);
		auto tree3 = [&](const Prob *t, unsigned s) {
			const unsigned b2 = s >> 2, b1 = (s >> 1) & 1, b0 = s & 1;
			return bp.bit(t[1], b2) + bp.bit(t[2 + b2], b1) + bp.bit(t[4 + (b2 << 1) + b1], b0);
		};
		for (unsigned ps = 0; ps < pos_states; ps++) {
			uint32_t *r = row[ps];
			for (unsigned s = 0; s < 8 && s + kLenMin <= max_len; s++)
				r[s + kLenMin] = c0 + tree3(m.low[ps], s);
			for (unsigned s = 0; s < 8 && s + 8 + kLenMin <= max_len; s++)
				r[s + 8 + kLenMin] = c10 + tree3(m.mid[ps], s);
		}
		if (max_len >= 16 + kLenMin) {
			// the 8-bit tree is shared by all position states: walk it once, pairs of leaves share 7 levels
			uint32_t *r0 = row[0];
			const unsigned leaves = max_len - (16 + kLenMin) + 1;
			for (unsigned s = 0; s < leaves; s += 2) {
				uint32_t pr = c11;
				unsigned node = (s >> 1) + 128;
				const unsigned parent = node;
				while (node >= 2) {
					pr += bp.bit(m.high[node >> 1], node & 1);
					node >>= 1;
				}
				r0[s + 16 + kLenMin] = pr + bp.zero(m.high[parent]);
				r0[s + 17 + kLenMin] = pr + bp.one(m.high[parent]);
			}
			for (unsigned ps = 1; ps < pos_states; ps++)
				memcpy(row[ps] + 16 + kLenMin, r0 + 16 + kLenMin, (leaves + (leaves & 1)) * sizeof(uint32_t));
		}
	}
};

struct LzmaModel {
	Prob is_match[kStates][kPosStatesMax];
	Prob is_rep[kStates], is_rep0[kStates], is_rep1[kStates], is_rep2[kStates];
	Prob is_rep0_long[kStates][kPosStatesMax];
	Prob slot[kLenToDistStates][kDistSlots];
	Prob near_footer[kNearDistances]; // reverse trees of distances 4..127, rooted at their slot's base
	Prob align[kAlignSize];
	LenModel match_len, rep_len;
	std::vector<Prob> literal; // 0x300 per context
	unsigned lc = 3, lp = 0, pb = 2;
	uint32_t lp_mask = 0;

	void reset(unsigned lc_, unsigned lp_, unsigned pb_)
	{
		lc = lc_;
		lp = lp_;
		pb = pb_;
		lp_mask = (0x100u << lp) - (0x100u >> lc);
		for (auto &r : is_match)
			for (Prob &p : r)
				p = kProbHalf;
		for (auto &r : is_rep0_long)
			for (Prob &p : r)
				p = kProbHalf;
		for (unsigned s = 0; s < kStates; s++)
			is_rep[s] = is_rep0[s] = is_rep1[s] = is_rep2[s] = kProbHalf;
		for (auto &r : slot)
			for (Prob &p : r)
				p = kProbHalf;
		for (Prob &p : near_footer)
			p = kProbHalf;
		for (Prob &p : align)
			p = kProbHalf;
		match_len.reset();
		rep_len.reset();
		literal.assign((size_t)0x300 << (lc + lp), kProbHalf);
	}
	inline Prob *literal_context(uint32_t pos, unsigned prev_byte)
	{
		return literal.data() + (size_t)3 * ((((pos << 8) + prev_byte) & lp_mask) << lc);
	}
	inline const Prob *literal_context(uint32_t pos, unsigned prev_byte) const
	{
		return literal.data() + (size_t)3 * ((((pos << 8) + prev_byte) & lp_mask) << lc);
	}
};

// price tables derived from the model; refreshed at the cadence the format's encoder uses
struct PriceTables {
	BitPrices bit;
	LenPrices match_len, rep_len;
	// distance prices, the four length contexts of one slot / one near distance adjacent: the parser prices a
	// pair for every length it covers with one 128-bit load (lengths 2, 3, 4 and "5 or more" differ only here)
	// Both live in ONE array of 16-byte rows -- row d for a near distance d, row kNearDistances + s for slot s -- so
	// that "the row of this pair" is an index the parser can compute for four pairs at once, without a branch.
	alignas(64) uint32_t dist_rows[kNearDistances + kDistSlots][kLenToDistStates];
	uint32_t (*const near_dist)[kLenToDistStates] = dist_rows;
	uint32_t (*const slot)[kLenToDistStates] = dist_rows + kNearDistances;
	alignas(64) uint32_t align[kAlignSize];

	// literal coded plainly: 8 tree levels, all node indices known from the symbol
	inline uint32_t literal(const Prob *ctx, unsigned sym) const
	{
		const unsigned s = sym | 0x100;
		return bit.bit(ctx[s >> 8], (s >> 7) & 1) + bit.bit(ctx[s >> 7], (s >> 6) & 1) + bit.bit(ctx[s >> 6], (s >> 5) & 1) +
		       bit.bit(ctx[s >> 5], (s >> 4) & 1) + bit.bit(ctx[s >> 4], (s >> 3) & 1) + bit.bit(ctx[s >> 3], (s >> 2) & 1) +
		       bit.bit(ctx[s >> 2], (s >> 1) & 1) + bit.bit(ctx[s >> 1], s & 1);
	}
	// literal after a match: while its bits agree with the byte at rep0 the tree is selected by that byte.  All eight
	// node indices are known from the two bytes: bit k (from the top) is coded in the matched half of the context
	// while the k bits above it agree -- `agree` has bit 8 - k set then -- and in the plain tree after that.
	inline uint32_t literal_matched(const Prob *ctx, unsigned sym, unsigned match_byte) const
	{
		unsigned x = (sym ^ match_byte) & 0xFF; // set bits: disagreements
		x |= x >> 1;
		x |= x >> 2;
		x |= x >> 4;                                   // ... and everything below the first one
		const unsigned agree = (~x & 0xFF) | 0x100;      // bit 8: nothing above the top bit; bit j: bits 7 .. j of the bytes agree
		const unsigned s = sym | 0x100, mb = match_byte << 1;
		uint32_t pr = 0;
#define LRZ_ML(k)                                                                                       \
	{                                                                                               \
		const unsigned offs = (agree << (k)) & 0x100;                                           \
		pr += bit.bit(ctx[offs + ((mb << (k)) & offs) + (s >> (8 - (k)))], (s >> (7 - (k))) & 1); \
	}
		LRZ_ML(0) LRZ_ML(1) LRZ_ML(2) LRZ_ML(3) LRZ_ML(4) LRZ_ML(5) LRZ_ML(6) LRZ_ML(7)
#undef LRZ_ML
		return pr;
	}

	void refresh_align(const LzmaModel &m)
	{
		for (unsigned v = 0; v < kAlignSize; v++) {
			uint32_t pr = 0;
			unsigned node = 1, s = v;
			for (unsigned k = 0; k < kAlignBits; k++) {
				const unsigned b = s & 1;
				s >>= 1;
				pr += bit.bit(m.align[node], b);
				node = (node << 1) | b;
			}
			align[v] = pr;
		}
	}
	// slot prices (+ the equiprobable middle bits of far distances) and the full price of every near distance
	void refresh_distances(const LzmaModel &m, unsigned slots_in_use)
	{
		uint32_t footer[kNearDistances];
		for (unsigned d = 4; d < kNearDistances; d++) {
			const unsigned sl = dist_slot(d), nb = (sl >> 1) - 1, base = (2 | (sl & 1)) << nb;
			uint32_t pr = 0;
			unsigned node = 1, s = d - base;
			for (unsigned k = 0; k < nb; k++) {
				const unsigned b = s & 1;
				s >>= 1;
				pr += bit.bit(m.near_footer[base + node], b);
				node = (node << 1) | b;
			}
			footer[d] = pr;
		}
		const unsigned slots = (slots_in_use + 1) & ~1u;
		for (unsigned ls = 0; ls < kLenToDistStates; ls++) {
			const Prob *t = m.slot[ls];
			for (unsigned sl = 0; sl < slots; sl++) {
				uint32_t pr = 0;
				unsigned node = sl + kDistSlots;
				while (node >= 2) {
					pr += bit.bit(t[node >> 1], node & 1);
					node >>= 1;
				}
				if (sl >= 14) // far distance: (sl/2 - 1) footer bits, all but the 4 align bits at one bit each
					pr += (uint32_t)((sl >> 1) - 1 - kAlignBits) << kPriceBitShift;
				slot[sl][ls] = pr;
			}
			for (unsigned d = 0; d < 4; d++)
				near_dist[d][ls] = slot[d][ls];
			for (unsigned d = 4; d < kNearDistances; d++)
				near_dist[d][ls] = slot[dist_slot(d)][ls] + footer[d];
		}
	}
	inline uint32_t distance(unsigned len_state, uint32_t d) const
	{
		return d < kNearDistances ? near_dist[d][len_state] : slot[dist_slot(d)][len_state] + align[d & (kAlignSize - 1)];
	}
};

} // namespace lrzgpu
