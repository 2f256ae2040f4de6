// filters_gpu.hip -- NOT part of liblrzgpu.so yet: kernels for the position-independent BCJ / delta converters
// (ARM, Thumb, PPC, SPARC, ARM64, IA-64 in place; the delta encoder out of place) with a self-test against the
// product's host converters (lrzip-next_amd/csrc/filters.cpp, themselves pinned to the reference's Bra.c / Delta.c).
// Every unit -- an aligned 32-bit word, a Thumb BL pair (pairs cannot overlap: the second halfword's 11111 is never a
// first halfword's 11110, and a conversion leaves those five bits alone), a 16-byte IA-64 bundle, a byte and the byte
// `dist` before it -- is converted from its own bytes and its own offset, so one thread per unit is exactly the
// serial scan.  HBM-bound: 2 B of traffic per block byte.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../lrzip-next_amd/csrc filters_gpu.hip ../../lrzip-next_amd/csrc/filters.cpp -o /tmp/filters_gpu && /tmp/filters_gpu
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "filters.h"

namespace {

__device__ __forceinline__ uint32_t conv(uint32_t v, uint32_t c, bool enc) { return enc ? v + c : v - c; }
__device__ __forceinline__ uint32_t bswap(uint32_t v) { return __builtin_bswap32(v); }

// one aligned 32-bit word per thread: ARM (flag 2), PPC (4), SPARC (5), ARM64 (7)
__global__ void __launch_bounds__(256) k_filter_word(uint32_t *__restrict__ w, size_t nwords, int flag, int enc_i)
{
	const bool enc = enc_i != 0;
	for (size_t k = blockIdx.x * (size_t)blockDim.x + threadIdx.x; k < nwords; k += (size_t)gridDim.x * blockDim.x) {
		const uint32_t i = (uint32_t)(k * 4);
		uint32_t v = w[k];
		if (flag == lrzgpu::FILTER_ARM) {
			if ((v >> 24) != 0xEB)
				continue;
			v = (conv(v, (i + 8) >> 2, enc) & 0x00FFFFFFu) | 0xEB000000u;
		} else if (flag == lrzgpu::FILTER_PPC) {
			uint32_t b = bswap(v);
			if ((b & 0xFC000003u) != 0x48000001u)
				continue;
			b = (conv(b, i, enc) & 0x03FFFFFFu) | 0x48000000u;
			v = bswap(b);
		} else if (flag == lrzgpu::FILTER_SPARC) {
			uint32_t b = bswap(v);
			const uint32_t top = b >> 22;
			if (top != 0x100 && top != 0x1FF)
				continue;
			b = conv(b << 2, i, enc) >> 2;
			b = (((0u - ((b >> 22) & 1)) << 22) & 0x3FFFFFFFu) | (b & 0x3FFFFFu) | 0x40000000u;
			v = bswap(b);
		} else { // ARM64
			const uint32_t flag20 = 1u << 20, mask = (1u << 24) - (flag20 << 1);
			if (((v - 0x94000000u) & 0xFC000000u) == 0) {
				v = (conv(v, i >> 2, enc) & 0x03FFFFFFu) | 0x94000000u;
			} else {
				v -= 0x90000000u;
				if ((v & 0x9F000000u) != 0)
					continue;
				v += flag20;
				if (v & mask)
					continue;
				uint32_t z = (v & 0xFFFFFFE0u) | (v >> 26);
				z = conv(z, (i >> 9) & ~7u, enc);
				v &= 0x1F;
				v |= 0x90000000u;
				v |= z << 26;
				v |= 0x00FFFFE0u & ((z & ((flag20 << 1) - 1)) - flag20);
			}
		}
		w[k] = v;
	}
}

// Thumb: one halfword index per thread; the pair (h, h + 1) is a BL iff 11110 / 11111
__global__ void __launch_bounds__(256) k_filter_armt(uint16_t *__restrict__ hw, size_t nhalf, int enc_i)
{
	const bool enc = enc_i != 0;
	for (size_t h = blockIdx.x * (size_t)blockDim.x + threadIdx.x; h + 1 < nhalf; h += (size_t)gridDim.x * blockDim.x) {
		const uint32_t a = hw[h], b = hw[h + 1];
		if ((a & 0xF800) != 0xF000 || (b & 0xF800) != 0xF800)
			continue;
		uint32_t v = ((a & 0x7FF) << 11) | (b & 0x7FF);
		v = conv(v, ((uint32_t)(h * 2) + 4) >> 1, enc);
		hw[h] = (uint16_t)(0xF000 | ((v >> 11) & 0x7FF));
		hw[h + 1] = (uint16_t)(0xF800 | (v & 0x7FF));
	}
}

// IA-64: one 16-byte bundle per thread
__global__ void __launch_bounds__(256) k_filter_ia64(uint8_t *__restrict__ d, size_t nbundles, int enc_i)
{
	const bool enc = enc_i != 0;
	const uint32_t slots_of = 0x334B0000u; // two bits per template pair (templates 0x10..0x1F): which slots may hold br.call
	for (size_t k = blockIdx.x * (size_t)blockDim.x + threadIdx.x; k < nbundles; k += (size_t)gridDim.x * blockDim.x) {
		uint8_t *p0 = d + k * 16;
		const unsigned t = p0[0] & 0x1F;
		// branch_slots[] of filters.cpp: {0 x16, 4,4,6,6,0,0,7,7,4,4,0,0,4,4,0,0}
		unsigned m = 0;
		if (t >= 16) {
			const unsigned q = (t - 16) >> 1; // pairs share an entry
			const unsigned tab[8] = {4, 6, 0, 7, 4, 0, 4, 0};
			m = tab[q];
		}
		(void)slots_of;
		for (unsigned slot = 0, bit = 5; slot < 3; slot++, bit += 41) {
			if (!((m >> slot) & 1))
				continue;
			uint8_t *p = p0 + (bit >> 3);
			const unsigned sh = bit & 7;
			uint64_t raw = 0;
			for (int j = 0; j < 6; j++)
				raw |= (uint64_t)p[j] << (8 * j);
			uint64_t ins = raw >> sh;
			if (((ins >> 37) & 0xF) != 0x5 || ((ins >> 9) & 0x7) != 0)
				continue;
			uint32_t v = (uint32_t)((ins >> 13) & 0xFFFFF) | ((uint32_t)(ins >> 36) & 1) << 20;
			v = conv(v << 4, (uint32_t)(k * 16), enc) >> 4;
			ins &= ~((uint64_t)0x8FFFFF << 13);
			ins |= (uint64_t)(v & 0xFFFFF) << 13;
			ins |= (uint64_t)(v & 0x100000) << (36 - 20);
			raw = (raw & (((uint64_t)1 << sh) - 1)) | (ins << sh);
			for (int j = 0; j < 6; j++)
				p[j] = (uint8_t)(raw >> (8 * j));
		}
	}
}

// delta encoder, out of place: dst[i] = src[i] - src[i - dist]
__global__ void __launch_bounds__(256) k_delta_encode(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, size_t n, unsigned dist)
{
	for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
		dst[i] = (uint8_t)(src[i] - (i >= dist ? src[i - dist] : 0));
}

#define CHECK(x)                                                                     \
	do {                                                                         \
		hipError_t e_ = (x);                                                 \
		if (e_ != hipSuccess) {                                              \
			fprintf(stderr, "HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); \
			exit(2);                                                     \
		}                                                                    \
	} while (0)

uint64_t rng_state = 0x9E3779B97F4A7C15ull;
uint32_t rnd()
{
	rng_state ^= rng_state << 13;
	rng_state ^= rng_state >> 7;
	rng_state ^= rng_state << 17;
	return (uint32_t)(rng_state >> 32);
}

// bytes with plenty of instructions of the kind `flag` converts
std::vector<uint8_t> code_like(int flag, size_t n)
{
	std::vector<uint8_t> d(n);
	for (auto &b : d)
		b = (uint8_t)rnd();
	using namespace lrzgpu;
	for (size_t i = 0; i + 16 <= n; i += 4) {
		const uint32_t r = rnd() % 10;
		if (flag == FILTER_ARM && r < 4)
			d[i + 3] = 0xEB;
		else if (flag == FILTER_PPC && r < 4) {
			d[i] = (uint8_t)(0x48 | (rnd() & 3));
			d[i + 3] = (uint8_t)((d[i + 3] & 0xFC) | 1);
		} else if (flag == FILTER_SPARC && r < 5) {
			if (r & 1) {
				d[i] = 0x40;
				d[i + 1] &= 0x3F;
			} else {
				d[i] = 0x7F;
				d[i + 1] |= 0xC0;
			}
		} else if (flag == FILTER_ARM64 && r < 6) {
			uint32_t v = r < 3 ? 0x94000000u | (rnd() & 0x03FFFFFFu)
					   : 0x90000000u | (rnd() & 3) << 29 | ((r == 3 ? rnd() & 0x7FFF : r == 4 ? 0x7FFFF ^ (rnd() & 0x7FFF) : rnd() & 0x7FFFF) << 5) | (rnd() & 31);
			memcpy(&d[i], &v, 4);
		} else if (flag == FILTER_ARMT && r < 4) {
			d[i + 1] = (uint8_t)(0xF0 | (rnd() & 7));
			d[i + 3] = (uint8_t)(0xF8 | (rnd() & 7));
		} else if (flag == FILTER_IA64 && (i & 15) == 0) {
			static const uint8_t tmpl[8] = {0x10, 0x12, 0x16, 0x18, 0x1C, 0x11, 0x13, 0x00};
			d[i] = (uint8_t)((d[i] & 0xE0) | tmpl[rnd() & 7]);
			for (unsigned slot = 0; slot < 3; slot++) {
				if (rnd() & 1)
					continue;
				const unsigned bit = 5 + 41 * slot;
				// opcode 5 at bits 37..40, btype 0 at bits 9..11 of the slot
				for (unsigned b = 37; b <= 40; b++) {
					const unsigned at = bit + b;
					d[i + (at >> 3)] = (uint8_t)((d[i + (at >> 3)] & ~(1u << (at & 7))) | ((((5u >> (b - 37)) & 1u)) << (at & 7)));
				}
				for (unsigned b = 9; b <= 11; b++) {
					const unsigned at = bit + b;
					d[i + (at >> 3)] &= (uint8_t)~(1u << (at & 7));
				}
			}
		}
	}
	return d;
}

} // namespace

int main()
{
	using namespace lrzgpu;
	int bad = 0;
	const int flags[] = {FILTER_ARM, FILTER_ARMT, FILTER_PPC, FILTER_SPARC, FILTER_IA64, FILTER_ARM64};
	const size_t sizes[] = {0, 3, 4, 17, 4096, 1000003, (size_t)8 << 20};
	uint8_t *dbuf = nullptr, *dbuf2 = nullptr;
	CHECK(hipMalloc(&dbuf, ((size_t)8 << 20) + 64));
	CHECK(hipMalloc(&dbuf2, ((size_t)8 << 20) + 64));
	for (int flag : flags)
		for (size_t n : sizes)
			for (int enc = 1; enc >= 0; enc--) {
				std::vector<uint8_t> data = code_like(flag, n), want = data, got(n);
				if (filter_block(flag, 0, want.data(), n, enc != 0) != 0) {
					printf("host filter refused flag %d\n", flag);
					return 1;
				}
				if (n)
					CHECK(hipMemcpy(dbuf, data.data(), n, hipMemcpyHostToDevice));
				const int grid = 1024;
				if (flag == FILTER_ARMT) {
					if (n / 2 >= 2)
						hipLaunchKernelGGL(k_filter_armt, dim3(grid), dim3(256), 0, 0, (uint16_t *)dbuf, n / 2, enc);
				} else if (flag == FILTER_IA64) {
					if (n / 16)
						hipLaunchKernelGGL(k_filter_ia64, dim3(grid), dim3(256), 0, 0, dbuf, n / 16, enc);
				} else if (n / 4)
					hipLaunchKernelGGL(k_filter_word, dim3(grid), dim3(256), 0, 0, (uint32_t *)dbuf, n / 4, flag, enc);
				CHECK(hipDeviceSynchronize());
				if (n)
					CHECK(hipMemcpy(got.data(), dbuf, n, hipMemcpyDeviceToHost));
				if (got != want) {
					size_t k = 0;
					while (k < n && got[k] == want[k])
						k++;
					printf("MISMATCH flag %d n %zu enc %d at %zu\n", flag, n, enc, k);
					bad++;
				}
			}
	for (unsigned dist : {1u, 2u, 4u, 16u, 48u, 256u})
		for (size_t n : sizes) {
			std::vector<uint8_t> data = code_like(FILTER_ARM, n), want = data, got(n);
			filter_block(FILTER_DELTA, (int)dist, want.data(), n, true);
			if (n) {
				CHECK(hipMemcpy(dbuf, data.data(), n, hipMemcpyHostToDevice));
				hipLaunchKernelGGL(k_delta_encode, dim3(1024), dim3(256), 0, 0, dbuf, dbuf2, n, dist);
				CHECK(hipDeviceSynchronize());
				CHECK(hipMemcpy(got.data(), dbuf2, n, hipMemcpyDeviceToHost));
			}
			if (got != want) {
				printf("MISMATCH delta %u n %zu\n", dist, n);
				bad++;
			}
		}
	printf("filters_gpu self-test: %d mismatches\n", bad);
	return bad ? 1 : 0;
}
