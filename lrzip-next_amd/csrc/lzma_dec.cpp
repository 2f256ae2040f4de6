// lzma_dec.cpp -- raw LZMA1 decoder written from the format (the mirror image of lzma_enc.cpp's
// coding decisions: same probability model layout, same state machine).  Behaviour reference:
// src/lzma/C/LzmaDec.c LzmaDec_DecodeReal.  The whole block is the dictionary (out[] is one
// contiguous buffer), so distances index straight into the output.
#include "lzma_dec.h"

#include <vector>

namespace lrzgpu {
namespace {

constexpr unsigned kNumBitModelTotalBits = 11, kBitModelTotal = 1u << kNumBitModelTotalBits, kNumMoveBits = 5;
constexpr uint32_t kTopValue = 1u << 24;
constexpr unsigned kNumStates = 12, kNumPosBitsMax = 4, kLenLowBits = 3, kLenHighBits = 8;
constexpr unsigned kNumLenToPosStates = 4, kNumPosSlotBits = 6, kStartPosModelIndex = 4, kEndPosModelIndex = 14;
constexpr unsigned kNumFullDistances = 1u << (kEndPosModelIndex >> 1), kNumAlignBits = 4, kMatchMinLen = 2;
typedef uint16_t Prob;

struct RangeDec {
	const uint8_t *p, *end;
	uint32_t range = 0xFFFFFFFFu, code = 0;
	bool bad = false;
	inline uint8_t next()
	{
		if (p == end) {
			bad = true;
			return 0;
		}
		return *p++;
	}
	void init()
	{
		if (next() != 0) // the first byte of a range-coded stream is always 0
			bad = true;
		for (int i = 0; i < 4; i++)
			code = (code << 8) | next();
	}
	inline void norm()
	{
		if (range < kTopValue) {
			range <<= 8;
			code = (code << 8) | next();
		}
	}
	inline unsigned bit(Prob *prob)
	{
		norm();
		const uint32_t bound = (range >> kNumBitModelTotalBits) * *prob;
		if (code < bound) {
			range = bound;
			*prob = (Prob)(*prob + ((kBitModelTotal - *prob) >> kNumMoveBits));
			return 0;
		}
		range -= bound;
		code -= bound;
		*prob = (Prob)(*prob - (*prob >> kNumMoveBits));
		return 1;
	}
	inline uint32_t direct(unsigned nbits)
	{
		uint32_t r = 0;
		while (nbits--) {
			norm();
			range >>= 1;
			const uint32_t t = (code - range) >> 31; // 1 if code < range
			code -= range & (t - 1);
			r = (r << 1) | (1 - t);
		}
		return r;
	}
	inline unsigned tree(Prob *probs, unsigned nbits)
	{
		unsigned m = 1;
		for (unsigned i = 0; i < nbits; i++)
			m = (m << 1) | bit(probs + m);
		return m - (1u << nbits);
	}
	inline unsigned tree_reverse(Prob *probs, unsigned nbits)
	{
		unsigned m = 1, sym = 0;
		for (unsigned i = 0; i < nbits; i++) {
			const unsigned b = bit(probs + m);
			m = (m << 1) | b;
			sym |= b << i;
		}
		return sym;
	}
};

struct LenDec {
	Prob choice, choice2;
	Prob low[1u << kNumPosBitsMax][1u << kLenLowBits];
	Prob mid[1u << kNumPosBitsMax][1u << kLenLowBits];
	Prob high[1u << kLenHighBits];
	void init()
	{
		choice = choice2 = kBitModelTotal / 2;
		for (auto &r : low)
			for (auto &p : r)
				p = kBitModelTotal / 2;
		for (auto &r : mid)
			for (auto &p : r)
				p = kBitModelTotal / 2;
		for (auto &p : high)
			p = kBitModelTotal / 2;
	}
	unsigned decode(RangeDec &rc, unsigned pos_state)
	{
		if (rc.bit(&choice) == 0)
			return rc.tree(low[pos_state], kLenLowBits);
		if (rc.bit(&choice2) == 0)
			return (1u << kLenLowBits) + rc.tree(mid[pos_state], kLenLowBits);
		return (2u << kLenLowBits) + rc.tree(high, kLenHighBits);
	}
};

} // namespace

int lzma_decode_block(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_len, unsigned lc, unsigned lp, unsigned pb)
{
	if (lc > 8 || lp > 4 || pb > 4)
		return -1;
	if (out_len == 0)
		return 0;
	RangeDec rc;
	rc.p = in;
	rc.end = in + in_len;
	rc.init();
	const Prob init = kBitModelTotal / 2;
	std::vector<Prob> lit((size_t)0x300 << (lc + lp), init);
	Prob is_match[kNumStates][1u << kNumPosBitsMax], is_rep[kNumStates], is_rep_g0[kNumStates], is_rep_g1[kNumStates],
		is_rep_g2[kNumStates], is_rep0_long[kNumStates][1u << kNumPosBitsMax];
	Prob pos_slot[kNumLenToPosStates][1u << kNumPosSlotBits], pos_dec[1 + kNumFullDistances - kEndPosModelIndex],
		align[1u << kNumAlignBits];
	for (auto &r : is_match)
		for (auto &p : r)
			p = init;
	for (auto &r : is_rep0_long)
		for (auto &p : r)
			p = init;
	for (unsigned s = 0; s < kNumStates; s++)
		is_rep[s] = is_rep_g0[s] = is_rep_g1[s] = is_rep_g2[s] = init;
	for (auto &r : pos_slot)
		for (auto &p : r)
			p = init;
	for (auto &p : pos_dec)
		p = init;
	for (auto &p : align)
		p = init;
	LenDec len_dec, rep_len_dec;
	len_dec.init();
	rep_len_dec.init();

	unsigned state = 0;
	uint32_t rep0 = 0, rep1 = 0, rep2 = 0, rep3 = 0; // distances - 1
	const unsigned pb_mask = (1u << pb) - 1, lp_mask = (1u << lp) - 1;
	size_t pos = 0;
	while (pos < out_len) {
		if (rc.bad)
			return -1;
		const unsigned pos_state = (unsigned)pos & pb_mask;
		if (rc.bit(&is_match[state][pos_state]) == 0) {
			const unsigned prev = pos ? out[pos - 1] : 0;
			Prob *probs = lit.data() + (size_t)0x300 * ((((unsigned)pos & lp_mask) << lc) + (prev >> (8 - lc)));
			unsigned sym = 1;
			if (state >= 7) { // after a match: the byte at rep0 steers the tree until the first mismatch
				if ((size_t)rep0 + 1 > pos)
					return -1;
				unsigned match_byte = out[pos - rep0 - 1];
				do {
					const unsigned mb = (match_byte >> 7) & 1;
					match_byte <<= 1;
					const unsigned b = rc.bit(probs + ((1 + mb) << 8) + sym);
					sym = (sym << 1) | b;
					if (mb != b)
						break;
				} while (sym < 0x100);
			}
			while (sym < 0x100)
				sym = (sym << 1) | rc.bit(probs + sym);
			out[pos++] = (uint8_t)sym;
			state = state < 4 ? 0 : state < 10 ? state - 3 : state - 6;
			continue;
		}
		unsigned len;
		if (rc.bit(&is_rep[state]) != 0) {
			if (pos == 0)
				return -1;
			if (rc.bit(&is_rep_g0[state]) == 0) {
				if (rc.bit(&is_rep0_long[state][pos_state]) == 0) { // short rep: one byte
					if ((size_t)rep0 + 1 > pos)
						return -1;
					state = state < 7 ? 9 : 11;
					out[pos] = out[pos - rep0 - 1];
					pos++;
					continue;
				}
			} else {
				uint32_t dist;
				if (rc.bit(&is_rep_g1[state]) == 0)
					dist = rep1;
				else {
					if (rc.bit(&is_rep_g2[state]) == 0)
						dist = rep2;
					else {
						dist = rep3;
						rep3 = rep2;
					}
					rep2 = rep1;
				}
				rep1 = rep0;
				rep0 = dist;
			}
			len = rep_len_dec.decode(rc, pos_state);
			state = state < 7 ? 8 : 11;
		} else {
			rep3 = rep2;
			rep2 = rep1;
			rep1 = rep0;
			len = len_dec.decode(rc, pos_state);
			state = state < 7 ? 7 : 10;
			const unsigned slot = rc.tree(pos_slot[len < kNumLenToPosStates ? len : kNumLenToPosStates - 1], kNumPosSlotBits);
			if (slot < kStartPosModelIndex)
				rep0 = slot;
			else {
				const unsigned nbits = (slot >> 1) - 1;
				rep0 = (2 | (slot & 1)) << nbits;
				if (slot < kEndPosModelIndex)
					rep0 += rc.tree_reverse(pos_dec + rep0 - slot, nbits); // LzmaDec.c's "- slot - 1 + 1" layout
				else {
					rep0 += rc.direct(nbits - kNumAlignBits) << kNumAlignBits;
					rep0 += rc.tree_reverse(align, kNumAlignBits);
					if (rep0 == 0xFFFFFFFFu)
						return -1; // end marker: lrzip-next never writes one
				}
			}
		}
		len += kMatchMinLen;
		if ((size_t)rep0 + 1 > pos || rc.bad)
			return -1;
		if (len > out_len - pos)
			return -1;
		const uint8_t *src = out + pos - rep0 - 1;
		for (unsigned k = 0; k < len; k++) // byte-wise: source and destination may overlap
			out[pos + k] = src[k];
		pos += len;
	}
	return rc.bad ? -1 : 0;
}

} // namespace lrzgpu
