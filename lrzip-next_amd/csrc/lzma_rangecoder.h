// lzma_rangecoder.h -- the LZMA binary range coder, encoder side (host).
//
// Format facts it implements (the decoder in lzma_dec.cpp is its inverse; reference behaviour:
// src/lzma/C/LzmaEnc.c:631-759): 32-bit range, 11-bit adaptive probabilities moved by 1/32 per coded
// bit, renormalisation one byte at a time when the range drops below 2^24, carries resolved through a
// held-back byte plus a run counter of 0xFF bytes, first output byte always 0, 5 flush bytes.
// Every operation is branch-free on the coded bit: the bit values of compressed data are as
// unpredictable as data gets, and a mispredicted branch per coded bit would cost more than the coding.
#pragma once
#include <cstddef>
#include <cstdint>

namespace lrzgpu {

typedef uint16_t Prob;
constexpr unsigned kProbBits = 11;
constexpr unsigned kProbOne = 1u << kProbBits; // probability 1.0
constexpr Prob kProbHalf = kProbOne >> 1;
constexpr unsigned kAdaptShift = 5;

struct RangeEncoder {
	uint64_t low = 0;          // 33 bits: bit 32 is a carry into bytes not yet released
	uint32_t range = 0xFFFFFFFFu;
	uint8_t held = 0;          // the byte a carry would still change
	uint64_t held_ff = 0;      // 0xFF bytes after it that a carry would turn into 0x00
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
	// move the top byte of `low` out of the window
	inline void shift()
	{
		const uint32_t lo = (uint32_t)low;
		const unsigned carry = (unsigned)(low >> 32);
		low = (uint64_t)(uint32_t)(lo << 8);
		if (lo >= 0xFF000000u && !carry) { // 0xFF and undecided: it joins the run
			held_ff++;
			return;
		}
		put((uint8_t)(held + carry));
		held = (uint8_t)(lo >> 24);
		for (; held_ff; held_ff--)
			put((uint8_t)(0xFF + carry));
	}
	inline void renorm()
	{
		if (range < (1u << 24)) {
			range <<= 8;
			shift();
		}
	}
	inline void encode(Prob *prob, unsigned bit)
	{
		const uint32_t p = *prob;
		const uint32_t bound = (range >> kProbBits) * p;
		const uint32_t one = 0u - bit; // all ones when the bit is 1
		low += bound & one;
		range = (bound & ~one) | ((range - bound) & one);
		const uint32_t up = (kProbOne - p) >> kAdaptShift, down = p >> kAdaptShift;
		*prob = (Prob)(p + (up & ~one) - (down & one));
		renorm();
	}
	inline void encode_direct(uint32_t value, unsigned nbits) // equiprobable bits, most significant first
	{
		while (nbits--) {
			range >>= 1;
			low += range & (0u - ((value >> nbits) & 1));
			renorm();
		}
	}
	// bit-tree, most significant bit first (literals, length-high, distance slots)
	template <unsigned NBITS> inline void encode_tree(Prob *probs, unsigned sym)
	{
		unsigned m = 1;
		for (unsigned k = NBITS; k--;) {
			const unsigned b = (sym >> k) & 1;
			encode(probs + m, b);
			m = (m << 1) | b;
		}
	}
	// bit-tree, least significant bit first (distance footers, align bits)
	inline void encode_tree_reverse(Prob *probs, unsigned nbits, unsigned sym)
	{
		unsigned m = 1;
		while (nbits--) {
			const unsigned b = sym & 1;
			sym >>= 1;
			encode(probs + m, b);
			m = (m << 1) | b;
		}
	}
	inline void finish()
	{
		for (int i = 0; i < 5; i++)
			shift();
	}
};

} // namespace lrzgpu
