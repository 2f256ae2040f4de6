// filters.cpp -- the executable-code and delta filters lrzip-next may run over a literal block before its back end
// and undo after it (src/stream.c:1587-1628 compress side, 1926-1990 decompress side; the converters themselves are
// the LZMA SDK's src/lzma/C/Bra.c, Bra86.c, Delta.c).  Host code, written from what the converters DO -- a branch
// instruction's relative target becomes absolute (relative to the start of the block: the reference always starts a
// block at pc = 0 with a fresh x86 state) -- one plain loop per instruction set; checked byte for byte against the
// reference's converters compiled unmodified into oracle/_ref (tests/test_filters_cpu.py).
//
// Position independence: ARM, ARM64, PPC, SPARC (one aligned 32-bit word each), Thumb (two halfwords that cannot
// overlap another pair), IA-64 (one 16-byte bundle) and the delta ENcoder convert every unit from its own bytes and
// its own offset only -- those are the GPU kernels of the next step; x86 carries a three-bit history of the bytes just
// passed, RISC-V steps over what it converts (4 or 8 bytes) and the delta DEcoder is a running sum.
#include "filters.h"

#include <cstring>

namespace lrzgpu {
namespace {

inline uint32_t le32(const uint8_t *p) { return (uint32_t)p[0] | (uint32_t)p[1] << 8 | (uint32_t)p[2] << 16 | (uint32_t)p[3] << 24; }
inline void put_le32(uint8_t *p, uint32_t v)
{
	p[0] = (uint8_t)v;
	p[1] = (uint8_t)(v >> 8);
	p[2] = (uint8_t)(v >> 16);
	p[3] = (uint8_t)(v >> 24);
}
inline uint32_t be32(const uint8_t *p) { return (uint32_t)p[0] << 24 | (uint32_t)p[1] << 16 | (uint32_t)p[2] << 8 | (uint32_t)p[3]; }
inline void put_be32(uint8_t *p, uint32_t v)
{
	p[0] = (uint8_t)(v >> 24);
	p[1] = (uint8_t)(v >> 16);
	p[2] = (uint8_t)(v >> 8);
	p[3] = (uint8_t)v;
}
inline uint32_t shift_by(uint32_t v, uint32_t c, bool enc) { return enc ? v + c : v - c; }

// ARM (A32): BL, condition "always": the top byte of the little-endian word is 0xEB; 24-bit word offset relative to
// the instruction + 8
void arm(uint8_t *d, size_t n, bool enc)
{
	for (size_t i = 0; i + 4 <= n; i += 4)
		if (d[i + 3] == 0xEB) {
			const uint32_t v = shift_by(le32(d + i), (uint32_t)(i + 8) >> 2, enc);
			put_le32(d + i, (v & 0x00FFFFFFu) | 0xEB000000u);
		}
}

// Thumb: BL as a pair of halfwords 11110 imm11 / 11111 imm11: 22-bit halfword offset relative to the pair + 4.
// A converted pair is stepped over as a whole (its second half cannot start another one: 11111 is not 11110).
void armt(uint8_t *d, size_t n, bool enc)
{
	n &= ~(size_t)1;
	for (size_t i = 0; i + 4 <= n; i += 2) {
		if ((d[i + 1] & 0xF8) != 0xF0 || (d[i + 3] & 0xF8) != 0xF8)
			continue;
		uint32_t v = ((uint32_t)(d[i + 1] & 7) << 19) | ((uint32_t)d[i] << 11) | ((uint32_t)(d[i + 3] & 7) << 8) | d[i + 2];
		v = shift_by(v, (uint32_t)(i + 4) >> 1, enc);
		d[i + 1] = (uint8_t)(0xF0 | ((v >> 19) & 7));
		d[i] = (uint8_t)(v >> 11);
		d[i + 3] = (uint8_t)(0xF8 | ((v >> 8) & 7));
		d[i + 2] = (uint8_t)v;
		i += 2;
	}
}

// PowerPC: bl (opcode 18, AA = 0, LK = 1) in a big-endian word; byte offset relative to the instruction
void ppc(uint8_t *d, size_t n, bool enc)
{
	for (size_t i = 0; i + 4 <= n; i += 4) {
		const uint32_t w = be32(d + i);
		if ((w & 0xFC000003u) != 0x48000001u)
			continue;
		const uint32_t v = shift_by(w, (uint32_t)i, enc);
		put_be32(d + i, (v & 0x03FFFFFFu) | 0x48000000u);
	}
}

// SPARC: call (01 disp30) whose 30-bit word displacement is a sign-extended 22-bit value (the top ten bits are
// 0100000000 or 0111111111); the converted value is brought back to that shape
void sparc(uint8_t *d, size_t n, bool enc)
{
	for (size_t i = 0; i + 4 <= n; i += 4) {
		const uint32_t top = ((uint32_t)d[i] << 2) | (d[i + 1] >> 6);
		if (top != 0x100 && top != 0x1FF)
			continue;
		uint32_t v = be32(d + i) << 2;
		v = shift_by(v, (uint32_t)i, enc) >> 2;
		v = (((0u - ((v >> 22) & 1)) << 22) & 0x3FFFFFFFu) | (v & 0x3FFFFFu) | 0x40000000u;
		put_be32(d + i, v);
	}
}

// ARM64: BL (100101 imm26: word offset relative to the instruction) and ADRP (1 immlo 10000 immhi Rd: page offset
// relative to the instruction's page).  ADRP is only converted when its 21-bit page offset lies in [-2^17, 2^17): the
// offset is biased by 2^17 into an 18-bit unsigned value, the page of the instruction is added (mod 2^18), the bias
// taken off again -- so that the decoder recognises exactly the same instructions.
void arm64(uint8_t *d, size_t n, bool enc)
{
	for (size_t i = 0; i + 4 <= n; i += 4) {
		uint32_t w = le32(d + i);
		if ((w & 0xFC000000u) == 0x94000000u) {
			w = (shift_by(w, (uint32_t)i >> 2, enc) & 0x03FFFFFFu) | 0x94000000u;
			put_le32(d + i, w);
			continue;
		}
		if ((w & 0x9F000000u) != 0x90000000u)
			continue;
		const uint32_t immhi = (w >> 5) & 0x7FFFFu, biased_hi = (immhi + 0x8000u) & 0x7FFFFu;
		if (biased_hi >> 16)
			continue; // page offset outside [-2^17, 2^17)
		uint32_t page_off = (biased_hi << 2) | ((w >> 29) & 3); // 18 bits, biased by 2^17
		page_off = shift_by(page_off, (uint32_t)i >> 12, enc) & 0x3FFFFu;
		const uint32_t new_hi = ((page_off >> 2) - 0x8000u) & 0x7FFFFu;
		put_le32(d + i, (w & 0x9F00001Fu) | ((page_off & 3) << 29) | (new_hi << 5));
	}
}

// IA-64: a 16-byte bundle = 5 template bits + three 41-bit slots; the template says which slots hold a branch unit
// instruction; br.call (opcode 5, btype 0) carries a 21-bit bundle offset (imm20b at bits 13..32, sign at 36)
void ia64(uint8_t *d, size_t n, bool enc)
{
	static const uint8_t branch_slots[32] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 4, 4, 6, 6, 0, 0, 7, 7, 4, 4, 0, 0, 4, 4, 0, 0};
	for (size_t i = 0; i + 16 <= n; i += 16) {
		const unsigned m = branch_slots[d[i] & 0x1F];
		for (unsigned slot = 0, bit = 5; slot < 3; slot++, bit += 41) {
			if (!((m >> slot) & 1))
				continue;
			uint8_t *p = d + i + (bit >> 3);
			const unsigned sh = bit & 7;
			uint64_t raw = 0;
			for (int k = 0; k < 6; k++)
				raw |= (uint64_t)p[k] << (8 * k);
			uint64_t ins = raw >> sh;
			if (((ins >> 37) & 0xF) != 0x5 || ((ins >> 9) & 0x7) != 0)
				continue;
			uint32_t v = (uint32_t)((ins >> 13) & 0xFFFFF) | ((uint32_t)(ins >> 36) & 1) << 20;
			v = shift_by(v << 4, (uint32_t)i, enc) >> 4;
			ins &= ~((uint64_t)0x8FFFFF << 13);
			ins |= (uint64_t)(v & 0xFFFFF) << 13;
			ins |= (uint64_t)(v & 0x100000) << (36 - 20);
			raw = (raw & (((uint64_t)1 << sh) - 1)) | (ins << sh);
			for (int k = 0; k < 6; k++)
				p[k] = (uint8_t)(raw >> (8 * k));
		}
	}
}

// x86: CALL / JMP rel32 (E8 / E9) whose top byte is 00 or FF -- a plausible code offset; `recent` remembers which of
// the last three bytes were E8/E9 themselves (an opcode byte inside the operand of another candidate makes both
// suspect: such candidates are skipped or tested on the byte the earlier one would have covered)
inline bool sign_byte(unsigned b) { return b == 0x00 || b == 0xFF; }
void x86(uint8_t *d, size_t n, bool enc)
{
	if (n < 5)
		return;
	const size_t limit = n - 4; // a candidate needs its four operand bytes
	unsigned recent = 0;
	size_t pos = 0;
	for (;;) {
		size_t p = pos;
		while (p < limit && (d[p] & 0xFE) != 0xE8)
			p++;
		const size_t gap = p - pos;
		pos = p;
		if (p >= limit)
			return;
		if (gap > 2)
			recent = 0;
		else {
			recent >>= gap;
			if (recent != 0 && (recent > 4 || recent == 3 || sign_byte(d[p + (recent >> 1) + 1]))) {
				recent = (recent >> 1) | 4;
				pos++;
				continue;
			}
		}
		if (sign_byte(d[p + 4])) {
			uint32_t v = le32(d + p + 1);
			const uint32_t cur = (uint32_t)pos + 5;
			pos += 5;
			v = shift_by(v, cur, enc);
			if (recent != 0) {
				const unsigned sh = (recent & 6) << 2;
				if (sign_byte((uint8_t)(v >> sh))) {
					v ^= ((uint32_t)0x100 << sh) - 1;
					v = shift_by(v, cur, enc);
				}
				recent = 0;
			}
			d[p + 1] = (uint8_t)v;
			d[p + 2] = (uint8_t)(v >> 8);
			d[p + 3] = (uint8_t)(v >> 16);
			d[p + 4] = (uint8_t)(0 - ((v >> 24) & 1));
		} else {
			recent = (recent >> 1) | 4;
			pos++;
		}
	}
}

// RISC-V (2-byte aligned scan over 16-bit parcels whose low seven bits are the JAL or the AUIPC opcode):
//  * JAL with rd = x1 or x5 (a call): the 20-bit byte offset, scattered over bits 12..31, becomes the absolute
//    address, stored most significant bits first in bytes 1 (high nibble), 2, 3;
//  * AUIPC rd, hi followed by a 32-bit instruction that reads rd (rs1 == rd, rd not x0 / x2): the pair's 32-bit
//    pc-relative offset becomes an absolute address stored big-endian in the second word, and the first word becomes
//    the marker "AUIPC x2" whose upper twenty bits carry the second instruction's lower twenty;
//  * an AUIPC x2 of the input that could be mistaken for that marker (bits 12, 13 set, top five bits not 0 / 2) is
//    escaped by trading fields with the word after it, so that the decoder meets an AUIPC whose rd is those five bits.
// The decoder tells the three apart by the same tests and undoes each.
inline uint32_t le16(const uint8_t *p) { return (uint32_t)p[0] | (uint32_t)p[1] << 8; }
inline bool rv_pair(uint32_t key, uint32_t next) { return (((next - 3) ^ (key << 8)) & 0xF8003u) == 0; } // 32-bit instruction whose rs1 is the AUIPC's rd
inline bool rv_marker_like(uint32_t key, uint32_t top5) { return (uint32_t)((key - 0x3108u) << 18) < (top5 & 0x1Du); }
void riscv(uint8_t *d, size_t n, bool enc)
{
	n &= ~(size_t)1;
	if (n <= 6)
		return;
	const size_t lim = n - 6;
	size_t i = 0;
	while (i < lim) {
		const uint32_t key = (le16(d + i) ^ 0x10u) + 1; // low 7 bits 0 <=> opcode 0x6F (JAL) or 0x17 (AUIPC); bits 7..11 = rd
		if (key & 0x77) {
			i += 2;
			continue;
		}
		uint8_t *p = d + i;
		const uint32_t a = le32(p);
		if (!(key & 8)) { // JAL
			if ((key - 0x100) & 0xD80) { // rd is neither x1 nor x5
				i += 2;
				continue;
			}
			if (enc) {
				uint32_t v = ((a & 0x80000000u) >> 11) | ((a & (0x3FFu << 21)) >> 20) | ((a & (1u << 20)) >> 9) | (a & (0xFFu << 12));
				v += (uint32_t)i;
				p[1] = (uint8_t)(((v >> 13) & 0xF0) | ((a >> 8) & 0xF));
				p[2] = (uint8_t)(v >> 9);
				p[3] = (uint8_t)(v >> 1);
			} else {
				uint32_t v = ((uint32_t)p[3] << 1) | ((uint32_t)p[2] << 9) | ((uint32_t)(p[1] & 0xF0) << 13);
				v -= (uint32_t)i;
				put_le32(p, (a & 0xFFF) | ((v << 11) & 0x80000000u) | ((v << 20) & (0x3FFu << 21)) | ((v << 9) & (1u << 20)) | (v & (0xFFu << 12)));
			}
			i += 4;
			continue;
		}
		// AUIPC
		const uint32_t next = le32(p + 4);
		if (key & 0xE80) { // rd is neither x0 nor x2
			if (!rv_pair(key, next)) {
				i += 6;
				continue;
			}
			if (enc) {
				put_le32(p, (next << 12) | 0x117u);
				put_be32(p + 4, (a & 0xFFFFF000u) + (uint32_t)((int32_t)next >> 20) + (uint32_t)i);
			} else { // an escaped AUIPC x2 of the original: give the fields back
				put_le32(p, (next << 12) | 0x117u);
				put_le32(p + 4, (a & 0xFFFFF000u) | (next >> 20));
			}
			i += 8;
		} else {
			const uint32_t top5 = a >> 27;
			if (!rv_marker_like(key, top5)) {
				i += 4;
				continue;
			}
			if (enc) { // looks like the marker: escape it
				put_le32(p, (top5 << 7) + 0x17u + (next & 0xFFFFF000u));
				put_le32(p + 4, (a >> 12) | (next << 20));
			} else { // the marker: the pair comes back from the absolute address
				const uint32_t off = be32(p + 4) - (uint32_t)i;
				put_le32(p, (top5 << 7) + 0x17u + ((off + 0x800u) & 0xFFFFF000u));
				put_le32(p + 4, (a >> 12) | (off << 20));
			}
			i += 8;
		}
	}
}

// delta: every byte minus the byte `dist` before it (zero history at the start of a block)
void delta_enc(uint8_t *d, size_t n, unsigned dist)
{
	for (size_t i = n; i-- > dist;)
		d[i] = (uint8_t)(d[i] - d[i - dist]);
}
void delta_dec(uint8_t *d, size_t n, unsigned dist)
{
	for (size_t i = dist; i < n; i++)
		d[i] = (uint8_t)(d[i] + d[i - dist]);
}

} // namespace

bool filter_supported(int flag, int delta)
{
	if (flag == FILTER_DELTA)
		return delta >= 1 && delta <= 256;
	return flag >= FILTER_X86 && flag <= FILTER_RISCV;
}

int filter_block(int flag, int delta, uint8_t *data, size_t n, bool encode)
{
	if (!filter_supported(flag, delta) || (n && !data))
		return -1;
	switch (flag) {
	case FILTER_X86: x86(data, n, encode); break;
	case FILTER_ARM: arm(data, n, encode); break;
	case FILTER_ARMT: armt(data, n, encode); break;
	case FILTER_PPC: ppc(data, n, encode); break;
	case FILTER_SPARC: sparc(data, n, encode); break;
	case FILTER_IA64: ia64(data, n, encode); break;
	case FILTER_ARM64: arm64(data, n, encode); break;
	case FILTER_RISCV: riscv(data, n, encode); break;
	case FILTER_DELTA:
		if (encode)
			delta_enc(data, n, (unsigned)delta);
		else
			delta_dec(data, n, (unsigned)delta);
		break;
	default: return -1;
	}
	return 0;
}

} // namespace lrzgpu
