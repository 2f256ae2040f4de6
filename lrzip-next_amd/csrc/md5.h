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
	void block(const uint8_t *p); // md5.cpp
	uint32_t h_[4];
	uint64_t total_;
	size_t fill_;
	uint8_t buf_[64];
};

} // namespace lrzgpu
