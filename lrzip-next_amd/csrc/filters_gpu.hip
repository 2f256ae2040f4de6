// filters_gpu.hip -- the executable-code and delta filters of the compress path on the device: a literal block is
// filtered where the scan left it (stream 1, in HBM) before its lz4-less back end sees it, so the finder reads the
// filtered bytes without the block ever going to the host and back (reference: compthread filters cti->s_buf in
// place, src/stream.c:1587-1628; converters src/lzma/C/Bra.c, Bra86.c, BraIA64.c, Delta.c -- restated from what they
// do in filters.cpp, which is pinned to the reference's own build; these kernels are checked against both).
//
// Encode direction only (the read side is host code).  Three shapes:
//   * ARM, PPC, SPARC, ARM64 (one aligned 32-bit word), Thumb (a BL halfword pair: pairs cannot overlap, the second
//     half's 11111 is never a first half's 11110, and a conversion leaves those bits alone), IA-64 (one 16-byte
//     bundle): every unit is converted from its own bytes and its own offset -- one thread per unit.
//   * delta: byte i minus byte i - dist of the ORIGINAL block -- out of place into scratch, copied back.
//   * x86 and RISC-V are scans with state: x86 remembers which of the last three bytes were E8/E9 and steps over the
//     operand of what it converts; RISC-V steps 2, 4, 6 or 8 bytes depending on what it finds.  Both decide from
//     ORIGINAL bytes only (everything a conversion writes lies behind the scan position), and both forget their
//     history quickly: after 7 bytes without an opcode byte the x86 scan is in its initial state at the next one;
//     after three parcels that are neither JAL nor AUIPC the RISC-V scan visits the next candidate whatever happened
//     before.  So: (1) compact the candidate positions (rocPRIM select), (2) one thread per run of candidates that lie
//     closer than that walks its run exactly like the serial scan and marks what is converted (and with which
//     history), (3) one thread per marked candidate converts it -- converted units never overlap.  The candidate
//     list holds every position if it must: a block dense in opcode bytes (not machine code) goes the same way, its
//     candidates forming fewer, longer runs (round 3 had a one-thread kernel for such blocks: seconds per block).
// Bound: HBM, ~3 B per block byte (count + select read the block, the word kernels read + write it).
#include <hip/hip_runtime.h>
#include <rocprim/device/device_select.hpp>
#include <rocprim/iterator/counting_iterator.hpp>

#include "common.h"
#include "filters.h"
#include "filters_gpu.h"

namespace lrzgpu {
namespace {

__device__ __forceinline__ uint32_t bswap(uint32_t v) { return __builtin_bswap32(v); }
__device__ __forceinline__ uint32_t ld32(const uint8_t *p) { return (uint32_t)p[0] | (uint32_t)p[1] << 8 | (uint32_t)p[2] << 16 | (uint32_t)p[3] << 24; }
__device__ __forceinline__ void st32(uint8_t *p, uint32_t v)
{
	p[0] = (uint8_t)v;
	p[1] = (uint8_t)(v >> 8);
	p[2] = (uint8_t)(v >> 16);
	p[3] = (uint8_t)(v >> 24);
}
__device__ __forceinline__ void st32be(uint8_t *p, uint32_t v)
{
	p[0] = (uint8_t)(v >> 24);
	p[1] = (uint8_t)(v >> 16);
	p[2] = (uint8_t)(v >> 8);
	p[3] = (uint8_t)v;
}

// ---- one aligned 32-bit word per thread: ARM (2), PPC (4), SPARC (5), ARM64 (7) --------------------------------------
__global__ void __launch_bounds__(256) k_filter_word(uint32_t *__restrict__ w, size_t nwords, int flag)
{
	for (size_t k = blockIdx.x * (size_t)blockDim.x + threadIdx.x; k < nwords; k += (size_t)gridDim.x * blockDim.x) {
		const uint32_t i = (uint32_t)(k * 4);
		uint32_t v = w[k];
		if (flag == FILTER_ARM) { // BL, condition "always": 24-bit word offset relative to the instruction + 8
			if ((v >> 24) != 0xEB)
				continue;
			v = ((v + ((i + 8) >> 2)) & 0x00FFFFFFu) | 0xEB000000u;
		} else if (flag == FILTER_PPC) { // bl (opcode 18, AA 0, LK 1), big-endian; byte offset relative to the instruction
			uint32_t b = bswap(v);
			if ((b & 0xFC000003u) != 0x48000001u)
				continue;
			b = ((b + i) & 0x03FFFFFFu) | 0x48000000u;
			v = bswap(b);
		} else if (flag == FILTER_SPARC) { // call whose disp30 is a sign-extended 22-bit value
			uint32_t b = bswap(v);
			const uint32_t top = b >> 22;
			if (top != 0x100 && top != 0x1FF)
				continue;
			b = ((b << 2) + i) >> 2;
			b = (((0u - ((b >> 22) & 1)) << 22) & 0x3FFFFFFFu) | (b & 0x3FFFFFu) | 0x40000000u;
			v = bswap(b);
		} else { // ARM64: BL imm26 (word offset); ADRP with a page offset in [-2^17, 2^17), biased to 18 unsigned bits
			if ((v & 0xFC000000u) == 0x94000000u) {
				v = ((v + (i >> 2)) & 0x03FFFFFFu) | 0x94000000u;
			} else {
				if ((v & 0x9F000000u) != 0x90000000u)
					continue;
				const uint32_t immhi = (v >> 5) & 0x7FFFFu, biased_hi = (immhi + 0x8000u) & 0x7FFFFu;
				if (biased_hi >> 16)
					continue;
				uint32_t page_off = (biased_hi << 2) | ((v >> 29) & 3);
				page_off = (page_off + (i >> 12)) & 0x3FFFFu;
				const uint32_t new_hi = ((page_off >> 2) - 0x8000u) & 0x7FFFFu;
				v = (v & 0x9F00001Fu) | ((page_off & 3) << 29) | (new_hi << 5);
			}
		}
		w[k] = v;
	}
}

// ---- Thumb: one halfword index per thread; (h, h + 1) is a BL iff 11110 imm11 / 11111 imm11 ---------------------------
__global__ void __launch_bounds__(256) k_filter_armt(uint16_t *__restrict__ hw, size_t nhalf)
{
	for (size_t h = blockIdx.x * (size_t)blockDim.x + threadIdx.x; h + 1 < nhalf; h += (size_t)gridDim.x * blockDim.x) {
		const uint32_t a = hw[h], b = hw[h + 1];
		if ((a & 0xF800) != 0xF000 || (b & 0xF800) != 0xF800)
			continue;
		uint32_t v = ((a & 0x7FF) << 11) | (b & 0x7FF);
		v += ((uint32_t)(h * 2) + 4) >> 1;
		hw[h] = (uint16_t)(0xF000 | ((v >> 11) & 0x7FF));
		hw[h + 1] = (uint16_t)(0xF800 | (v & 0x7FF));
	}
}

// ---- IA-64: one 16-byte bundle per thread; the template says which slots hold a branch-unit instruction ---------------
__global__ void __launch_bounds__(256) k_filter_ia64(uint8_t *__restrict__ d, size_t nbundles)
{
	for (size_t k = blockIdx.x * (size_t)blockDim.x + threadIdx.x; k < nbundles; k += (size_t)gridDim.x * blockDim.x) {
		uint8_t *p0 = d + k * 16;
		const unsigned t = p0[0] & 0x1F;
		if (t < 16)
			continue;
		// templates 0x10..0x1F in pairs: slot masks 4, 6, 0, 7, 4, 0, 4, 0 (three bits each)
		const uint32_t slot_masks = 04u | (06u << 3) | (00u << 6) | (07u << 9) | (04u << 12) | (00u << 15) | (04u << 18) | (00u << 21);
		const unsigned m = (slot_masks >> (3 * ((t - 16) >> 1))) & 7;
		for (unsigned slot = 0, bit = 5; slot < 3; slot++, bit += 41) {
			if (!((m >> slot) & 1))
				continue;
			uint8_t *p = p0 + (bit >> 3);
			const unsigned sh = bit & 7;
			uint64_t raw = 0;
			for (int j = 0; j < 6; j++)
				raw |= (uint64_t)p[j] << (8 * j);
			uint64_t ins = raw >> sh;
			if (((ins >> 37) & 0xF) != 0x5 || ((ins >> 9) & 0x7) != 0) // br.call: opcode 5, btype 0
				continue;
			uint32_t v = (uint32_t)((ins >> 13) & 0xFFFFF) | ((uint32_t)(ins >> 36) & 1) << 20;
			v = ((v << 4) + (uint32_t)(k * 16)) >> 4;
			ins &= ~((uint64_t)0x8FFFFF << 13);
			ins |= (uint64_t)(v & 0xFFFFF) << 13;
			ins |= (uint64_t)(v & 0x100000) << (36 - 20);
			raw = (raw & (((uint64_t)1 << sh) - 1)) | (ins << sh);
			for (int j = 0; j < 6; j++)
				p[j] = (uint8_t)(raw >> (8 * j));
		}
	}
}

// ---- delta encoder, out of place: dst[i] = src[i] - src[i - dist] (zero history) -----------------------------------------
__global__ void __launch_bounds__(256) k_delta_encode(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, size_t n, unsigned dist)
{
	for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
		dst[i] = (uint8_t)(src[i] - (i >= dist ? src[i - dist] : 0));
}

// ---- x86 ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool sign_byte(unsigned b) { return b == 0x00 || b == 0xFF; }
struct IsX86Opcode {
	const uint8_t *d;
	__device__ __forceinline__ bool operator()(uint32_t p) const { return (d[p] & 0xFE) == 0xE8; }
};
// RISC-V: a 16-bit parcel at an even offset whose low seven bits are the JAL (0x6F) or the AUIPC (0x17) opcode
__device__ __forceinline__ uint32_t rv_key(const uint8_t *p) { return (((uint32_t)p[0] | (uint32_t)p[1] << 8) ^ 0x10u) + 1; }
struct IsRvCandidate {
	const uint8_t *d;
	__device__ __forceinline__ bool operator()(uint32_t parcel) const { return (rv_key(d + 2 * (size_t)parcel) & 0x77) == 0; }
};

// candidates among items [0, n_items): `mode` 0 = x86 opcode bytes, 1 = RISC-V parcels
__global__ void __launch_bounds__(256) k_count_candidates(const uint8_t *__restrict__ d, uint32_t n_items, int mode, unsigned long long *total)
{
	unsigned mine = 0;
	for (size_t k = blockIdx.x * (size_t)blockDim.x + threadIdx.x; k < n_items; k += (size_t)gridDim.x * blockDim.x)
		mine += mode == 0 ? (unsigned)((d[k] & 0xFE) == 0xE8) : (unsigned)((rv_key(d + 2 * k) & 0x77) == 0);
	for (int o = 32; o; o >>= 1)
		mine += __shfl_down(mine, o);
	__shared__ unsigned part[4];
	if ((threadIdx.x & 63) == 0)
		part[threadIdx.x >> 6] = mine;
	__syncthreads();
	if (threadIdx.x == 0)
		atomicAdd(total, (unsigned long long)part[0] + part[1] + part[2] + part[3]);
}

// x86: pos[] = the opcode bytes below `limit` in order.  The thread of a run's first candidate (no opcode byte in the
// seven bytes before it: whatever the scan did there, it reaches this byte with an empty history) walks the run.
// mark[k] = 0 not converted, 0x80 | history converted with that three-bit history.
__global__ void __launch_bounds__(256) k_x86_resolve(const uint8_t *__restrict__ d, const uint32_t *__restrict__ pos, const int *__restrict__ n_pos,
						     uint8_t *__restrict__ mark)
{
	const uint32_t n = (uint32_t)*n_pos;
	for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
		if (i != 0 && pos[i] - pos[i - 1] < 8)
			continue;
		unsigned recent = 0;
		uint32_t scan = pos[i]; // where the scan resumed before this candidate (only its distance matters)
		for (uint32_t k = (uint32_t)i;; k++) {
			const uint32_t p = pos[k];
			bool skip = false;
			if (k != i) {
				if (p < scan) { // inside the operand of a converted candidate: never looked at
					mark[k] = 0;
					skip = true;
				} else {
					const uint32_t gap = p - scan;
					if (gap > 2)
						recent = 0;
					else {
						recent >>= gap;
						if (recent != 0 && (recent > 4 || recent == 3 || sign_byte(d[p + (recent >> 1) + 1]))) {
							recent = (recent >> 1) | 4;
							scan = p + 1;
							mark[k] = 0;
							skip = true;
						}
					}
				}
			}
			if (!skip) {
				if (sign_byte(d[p + 4])) {
					mark[k] = (uint8_t)(0x80 | recent);
					scan = p + 5;
					recent = 0;
				} else {
					mark[k] = 0;
					recent = (recent >> 1) | 4;
					scan = p + 1;
				}
			}
			if (k + 1 >= n || pos[k + 1] - p >= 8)
				break;
		}
	}
}
__global__ void __launch_bounds__(256) k_x86_apply(uint8_t *__restrict__ d, const uint32_t *__restrict__ pos, const int *__restrict__ n_pos,
						   const uint8_t *__restrict__ mark)
{
	const uint32_t n = (uint32_t)*n_pos;
	for (size_t k = blockIdx.x * (size_t)blockDim.x + threadIdx.x; k < n; k += (size_t)gridDim.x * blockDim.x) {
		const unsigned m = mark[k];
		if (!m)
			continue;
		const uint32_t p = pos[k], cur = p + 5;
		const unsigned recent = m & 7;
		uint32_t v = ld32(d + p + 1) + cur;
		if (recent != 0) {
			const unsigned sh = (recent & 6) << 2;
			if (sign_byte((uint8_t)(v >> sh))) {
				v ^= ((uint32_t)0x100 << sh) - 1;
				v += cur;
			}
		}
		d[p + 1] = (uint8_t)v;
		d[p + 2] = (uint8_t)(v >> 8);
		d[p + 3] = (uint8_t)(v >> 16);
		d[p + 4] = (uint8_t)(0 - ((v >> 24) & 1));
	}
}

// ---- RISC-V ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool rv_pair(uint32_t key, uint32_t next) { return (((next - 3) ^ (key << 8)) & 0xF8003u) == 0; }
__device__ __forceinline__ bool rv_marker_like(uint32_t key, uint32_t top5) { return (uint32_t)((key - 0x3108u) << 18) < (top5 & 0x1Du); }
enum { RV_NONE = 0, RV_JAL = 1, RV_PAIR = 2, RV_ESCAPE = 3 };
// what the scan does at a candidate it visits: the conversion (RV_*) and how far it steps
__device__ __forceinline__ unsigned rv_decide(const uint8_t *p, unsigned *step)
{
	const uint32_t key = rv_key(p);
	if (!(key & 8)) { // JAL
		if ((key - 0x100) & 0xD80) { // rd is neither x1 nor x5
			*step = 2;
			return RV_NONE;
		}
		*step = 4;
		return RV_JAL;
	}
	const uint32_t next = ld32(p + 4);
	if (key & 0xE80) { // AUIPC, rd neither x0 nor x2
		if (!rv_pair(key, next)) {
			*step = 6;
			return RV_NONE;
		}
		*step = 8;
		return RV_PAIR;
	}
	if (!rv_marker_like(key, ld32(p) >> 27)) {
		*step = 4;
		return RV_NONE;
	}
	*step = 8;
	return RV_ESCAPE;
}
__device__ __forceinline__ void rv_convert(uint8_t *p, uint32_t i, unsigned what)
{
	const uint32_t a = ld32(p);
	if (what == RV_JAL) {
		uint32_t v = ((a & 0x80000000u) >> 11) | ((a & (0x3FFu << 21)) >> 20) | ((a & (1u << 20)) >> 9) | (a & (0xFFu << 12));
		v += i;
		p[1] = (uint8_t)(((v >> 13) & 0xF0) | ((a >> 8) & 0xF));
		p[2] = (uint8_t)(v >> 9);
		p[3] = (uint8_t)(v >> 1);
		return;
	}
	const uint32_t next = ld32(p + 4);
	if (what == RV_PAIR) {
		st32(p, (next << 12) | 0x117u);
		st32be(p + 4, (a & 0xFFFFF000u) + (uint32_t)((int32_t)next >> 20) + i);
	} else {
		st32(p, ((a >> 27) << 7) + 0x17u + (next & 0xFFFFF000u));
		st32(p + 4, (a >> 12) | (next << 20));
	}
}
// pos[] = candidate PARCEL indices (byte offset / 2) below the scan's limit, in order.  A candidate with no other
// candidate among the three parcels before it is visited by the scan whatever came earlier (steps are at most 8
// bytes, and non-candidates step 2): its thread walks the run.  mark[k] = RV_*.
__global__ void __launch_bounds__(256) k_rv_resolve(const uint8_t *__restrict__ d, const uint32_t *__restrict__ pos, const int *__restrict__ n_pos,
						    uint8_t *__restrict__ mark)
{
	const uint32_t n = (uint32_t)*n_pos;
	for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
		if (i != 0 && pos[i] - pos[i - 1] < 4)
			continue;
		uint32_t cur = pos[i]; // parcel the scan stands on
		for (uint32_t k = (uint32_t)i;; k++) {
			const uint32_t q = pos[k];
			if (q < cur) {
				mark[k] = RV_NONE; // stepped over
			} else { // the first candidate at or after the scan position: non-candidates in between step one parcel
				unsigned step;
				const unsigned what = rv_decide(d + 2 * (size_t)q, &step);
				mark[k] = (uint8_t)what;
				cur = q + step / 2;
			}
			if (k + 1 >= n || pos[k + 1] - q >= 4)
				break;
		}
	}
}
__global__ void __launch_bounds__(256) k_rv_apply(uint8_t *__restrict__ d, const uint32_t *__restrict__ pos, const int *__restrict__ n_pos,
						  const uint8_t *__restrict__ mark)
{
	const uint32_t n = (uint32_t)*n_pos;
	for (size_t k = blockIdx.x * (size_t)blockDim.x + threadIdx.x; k < n; k += (size_t)gridDim.x * blockDim.x)
		if (mark[k])
			rv_convert(d + 2 * (size_t)pos[k], 2 * pos[k], mark[k]);
}

unsigned grid_for(size_t items)
{
	size_t g = (items + 255) / 256;
	return (unsigned)(g < 1 ? 1 : (g > 8192 ? 8192 : g));
}

} // namespace

size_t filter_scratch_bytes(int flag, size_t n)
{
	if (flag == FILTER_DELTA)
		return n + 256;
	if (flag != FILTER_X86 && flag != FILTER_RISCV)
		return 0;
	// counter + count, the candidate list (every position may be one: a block dense in opcode bytes -- not machine
	// code -- is walked by the same kernels, its candidates just form few long runs), marks, select's temp
	const size_t cap = n + 1024;
	size_t temp = 0, temp2 = 0;
	rocprim::counting_iterator<uint32_t> it(0);
	const int items = (int)(n > 0x7FFFFFFF ? 0x7FFFFFFF : n);
	(void)rocprim::select(nullptr, temp, it, (uint32_t *)nullptr, (int *)nullptr, (size_t)items, IsX86Opcode{nullptr});
	(void)rocprim::select(nullptr, temp2, it, (uint32_t *)nullptr, (int *)nullptr, (size_t)items, IsRvCandidate{nullptr});
	if (temp2 > temp)
		temp = temp2;
	return 256 + ((cap * 4 + 255) & ~(size_t)255) + ((cap + 255) & ~(size_t)255) + temp + 4096;
}

int filter_block_device(int flag, int delta, uint8_t *d, size_t n, uint8_t *scratch, size_t scratch_bytes, hipStream_t s)
{
	if (!filter_supported(flag, delta) || (n && !d) || n >= 0xFFFFFFF0u)
		return -1;
	if (n == 0)
		return 0;
	if (((uintptr_t)d & 3) != 0)
		return -1; // blocks start at multiples of the (page-rounded) block size in an aligned buffer
	if (scratch_bytes < filter_scratch_bytes(flag, n))
		return -2;
	switch (flag) {
	case FILTER_ARM:
	case FILTER_PPC:
	case FILTER_SPARC:
	case FILTER_ARM64:
		if (n >= 4)
			hipLaunchKernelGGL(k_filter_word, dim3(grid_for(n / 4)), dim3(256), 0, s, (uint32_t *)d, n / 4, flag);
		break;
	case FILTER_ARMT:
		if (n >= 4)
			hipLaunchKernelGGL(k_filter_armt, dim3(grid_for(n / 2)), dim3(256), 0, s, (uint16_t *)d, n / 2);
		break;
	case FILTER_IA64:
		if (n >= 16)
			hipLaunchKernelGGL(k_filter_ia64, dim3(grid_for(n / 16)), dim3(256), 0, s, d, n / 16);
		break;
	case FILTER_DELTA:
		hipLaunchKernelGGL(k_delta_encode, dim3(grid_for(n)), dim3(256), 0, s, (const uint8_t *)d, scratch, n, (unsigned)delta);
		if (hipMemcpyAsync(d, scratch, n, hipMemcpyDeviceToDevice, s) != hipSuccess)
			return -3;
		break;
	case FILTER_X86:
	case FILTER_RISCV: {
		const bool x86 = flag == FILTER_X86;
		size_t n_items; // positions the scan may stop at
		if (x86) {
			if (n < 5)
				return 0;
			n_items = n - 4;
		} else {
			const size_t ne = n & ~(size_t)1;
			if (ne <= 6)
				return 0;
			n_items = (ne - 6) / 2;
		}
		const size_t cap = n + 1024;
		unsigned long long *d_total = (unsigned long long *)scratch;
		int *d_count = (int *)(scratch + 64);
		uint32_t *d_pos = (uint32_t *)(scratch + 256);
		uint8_t *d_mark = scratch + 256 + ((cap * 4 + 255) & ~(size_t)255);
		uint8_t *d_temp = d_mark + ((cap + 255) & ~(size_t)255); // 256-byte aligned like the scratch itself: rocPRIM lays its
		                                                         // scan state out relative to this pointer
		size_t temp = scratch_bytes - (size_t)(d_temp - scratch);
		if (hipMemsetAsync(scratch, 0, 256, s) != hipSuccess)
			return -3;
		hipLaunchKernelGGL(k_count_candidates, dim3(grid_for(n_items)), dim3(256), 0, s, (const uint8_t *)d, (uint32_t)n_items, x86 ? 0 : 1, d_total);
		unsigned long long total = 0;
		if (d2h_pageable(&total, d_total, sizeof(total), s) != hipSuccess)
			return -3;
		if (total == 0)
			return 0;
		if (total > cap)
			return -2; // (cannot happen: there are no more candidates than positions)
		rocprim::counting_iterator<uint32_t> it(0);
		hipError_t e;
		if (x86)
			e = rocprim::select(d_temp, temp, it, d_pos, d_count, (size_t)n_items, IsX86Opcode{d}, s);
		else
			e = rocprim::select(d_temp, temp, it, d_pos, d_count, (size_t)n_items, IsRvCandidate{d}, s);
		if (e != hipSuccess)
			return -3;
		const unsigned g = grid_for((size_t)total);
		if (x86) {
			hipLaunchKernelGGL(k_x86_resolve, dim3(g), dim3(256), 0, s, (const uint8_t *)d, (const uint32_t *)d_pos, (const int *)d_count, d_mark);
			hipLaunchKernelGGL(k_x86_apply, dim3(g), dim3(256), 0, s, d, (const uint32_t *)d_pos, (const int *)d_count, (const uint8_t *)d_mark);
		} else {
			hipLaunchKernelGGL(k_rv_resolve, dim3(g), dim3(256), 0, s, (const uint8_t *)d, (const uint32_t *)d_pos, (const int *)d_count, d_mark);
			hipLaunchKernelGGL(k_rv_apply, dim3(g), dim3(256), 0, s, d, (const uint32_t *)d_pos, (const int *)d_count, (const uint8_t *)d_mark);
		}
		break;
	}
	default: return -1;
	}
	return hipGetLastError() == hipSuccess ? 0 : -3;
}

} // namespace lrzgpu
