// md5.h -- RFC 1321 digest for the trailing whole-file hash (reference: libgcrypt GCRY_MD_MD5 via
// src/rzip.c:571-573, 1195-1219; hash_code 1 is the reference default, src/main.c:789).
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstring>

namespace lrzgpu {

class Md5 {
      public:
	Md5() { reset(); }
	void reset()
	{
		h_[0] = 0x67452301u;
		h_[1] = 0xefcdab89u;
		h_[2] = 0x98badcfeu;
		h_[3] = 0x10325476u;
		total_ = 0;
		fill_ = 0;
	}
	void update(const uint8_t *p, size_t n)
	{
		total_ += n;
		if (fill_) {
			size_t k = 64 - fill_;
			if (k > n)
				k = n;
			memcpy(buf_ + fill_, p, k);
			fill_ += k;
			p += k;
			n -= k;
			if (fill_ < 64)
				return;
			block(buf_);
			fill_ = 0;
		}
		for (; n >= 64; p += 64, n -= 64)
			block(p);
		if (n) {
			memcpy(buf_, p, n);
			fill_ = n;
		}
	}
	void finish(uint8_t out[16])
	{
		uint64_t bits = total_ * 8;
		uint8_t pad[72] = {0x80};
		size_t padn = fill_ < 56 ? 56 - fill_ : 120 - fill_;
		for (int i = 0; i < 8; i++)
			pad[padn + i] = (uint8_t)(bits >> (8 * i));
		update(pad, padn + 8);
		for (int w = 0; w < 4; w++)
			for (int i = 0; i < 4; i++)
				out[4 * w + i] = (uint8_t)(h_[w] >> (8 * i));
	}

      private:
	static inline uint32_t rol(uint32_t v, int s) { return (v << s) | (v >> (32 - s)); }
	void block(const uint8_t *p)
	{
		static const uint32_t K[64] = {
			0xd76aa478, 0xe8c7b756, 0x242070db, 0xc1bdceee, 0xf57c0faf, 0x4787c62a, 0xa8304613, 0xfd469501, 0x698098d8, 0x8b44f7af,
			0xffff5bb1, 0x895cd7be, 0x6b901122, 0xfd987193, 0xa679438e, 0x49b40821, 0xf61e2562, 0xc040b340, 0x265e5a51, 0xe9b6c7aa,
			0xd62f105d, 0x02441453, 0xd8a1e681, 0xe7d3fbc8, 0x21e1cde6, 0xc33707d6, 0xf4d50d87, 0x455a14ed, 0xa9e3e905, 0xfcefa3f8,
			0x676f02d9, 0x8d2a4c8a, 0xfffa3942, 0x8771f681, 0x6d9d6122, 0xfde5380c, 0xa4beea44, 0x4bdecfa9, 0xf6bb4b60, 0xbebfbc70,
			0x289b7ec6, 0xeaa127fa, 0xd4ef3085, 0x04881d05, 0xd9d4d039, 0xe6db99e5, 0x1fa27cf8, 0xc4ac5665, 0xf4292244, 0x432aff97,
			0xab9423a7, 0xfc93a039, 0x655b59c3, 0x8f0ccc92, 0xffeff47d, 0x85845dd1, 0x6fa87e4f, 0xfe2ce6e0, 0xa3014314, 0x4e0811a1,
			0xf7537e82, 0xbd3af235, 0x2ad7d2bb, 0xeb86d391};
		static const int S[4][4] = {{7, 12, 17, 22}, {5, 9, 14, 20}, {4, 11, 16, 23}, {6, 10, 15, 21}};
		uint32_t w[16];
		for (int i = 0; i < 16; i++)
			w[i] = (uint32_t)p[4 * i] | ((uint32_t)p[4 * i + 1] << 8) | ((uint32_t)p[4 * i + 2] << 16) | ((uint32_t)p[4 * i + 3] << 24);
		uint32_t a = h_[0], b = h_[1], c = h_[2], d = h_[3];
		for (int i = 0; i < 64; i++) {
			uint32_t f;
			int g;
			switch (i >> 4) {
			case 0: f = (b & c) | (~b & d); g = i; break;
			case 1: f = (d & b) | (~d & c); g = (5 * i + 1) & 15; break;
			case 2: f = b ^ c ^ d; g = (3 * i + 5) & 15; break;
			default: f = c ^ (b | ~d); g = (7 * i) & 15; break;
			}
			uint32_t t = d;
			d = c;
			c = b;
			b = b + rol(a + f + K[i] + w[g], S[i >> 4][i & 3]);
			a = t;
		}
		h_[0] += a;
		h_[1] += b;
		h_[2] += c;
		h_[3] += d;
	}
	uint32_t h_[4];
	uint64_t total_;
	size_t fill_;
	uint8_t buf_[64];
};

} // namespace lrzgpu
