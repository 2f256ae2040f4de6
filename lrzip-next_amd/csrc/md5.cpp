// md5.cpp -- the MD5 compression function (RFC 1321), see md5.h.  Its own translation unit so that it is
// built by the host compiler that schedules the dependency chain best (csrc/Makefile).
#include "md5.h"

namespace lrzgpu {

static inline uint32_t rol(uint32_t v, int s) { return (v << s) | (v >> (32 - s)); }

// One 64-byte block, fully unrolled.  MD5 is one serial dependency chain (each step needs the previous
// step's result), so the whole-file digest runs at the latency of that chain: everything that does not
// depend on the newest word -- message word + constant + the oldest state word, and the half of the
// boolean function that only reads older words -- is computed beside the chain, and the two halves of
// F and G are ADDED (their set bits are disjoint) so that the chain is and/add/add/rotate/add.
void Md5::block(const uint8_t *p)
{
	uint32_t w[16];
	memcpy(w, p, 64); // little-endian host (x86-64)
	uint32_t a = h_[0], b = h_[1], c = h_[2], d = h_[3];
#define MD5_F(a, b, c, d, k, s, t) \
	a += w[k] + t + (d & ~b);      \
	a += (b & c);                  \
	a = rol(a, s) + b;
#define MD5_G(a, b, c, d, k, s, t) \
	a += w[k] + t + (c & ~d);      \
	a += (b & d);                  \
	a = rol(a, s) + b;
#define MD5_H(a, b, c, d, k, s, t) \
	a += w[k] + t;                 \
	a += b ^ (c ^ d);              \
	a = rol(a, s) + b;
#define MD5_I(a, b, c, d, k, s, t) \
	a += w[k] + t;                 \
	a += c ^ (b | ~d);             \
	a = rol(a, s) + b;
	MD5_F(a, b, c, d, 0, 7, 0xd76aa478u) MD5_F(d, a, b, c, 1, 12, 0xe8c7b756u) MD5_F(c, d, a, b, 2, 17, 0x242070dbu) MD5_F(b, c, d, a, 3, 22, 0xc1bdceeeu)
	MD5_F(a, b, c, d, 4, 7, 0xf57c0fafu) MD5_F(d, a, b, c, 5, 12, 0x4787c62au) MD5_F(c, d, a, b, 6, 17, 0xa8304613u) MD5_F(b, c, d, a, 7, 22, 0xfd469501u)
	MD5_F(a, b, c, d, 8, 7, 0x698098d8u) MD5_F(d, a, b, c, 9, 12, 0x8b44f7afu) MD5_F(c, d, a, b, 10, 17, 0xffff5bb1u) MD5_F(b, c, d, a, 11, 22, 0x895cd7beu)
	MD5_F(a, b, c, d, 12, 7, 0x6b901122u) MD5_F(d, a, b, c, 13, 12, 0xfd987193u) MD5_F(c, d, a, b, 14, 17, 0xa679438eu) MD5_F(b, c, d, a, 15, 22, 0x49b40821u)
	MD5_G(a, b, c, d, 1, 5, 0xf61e2562u) MD5_G(d, a, b, c, 6, 9, 0xc040b340u) MD5_G(c, d, a, b, 11, 14, 0x265e5a51u) MD5_G(b, c, d, a, 0, 20, 0xe9b6c7aau)
	MD5_G(a, b, c, d, 5, 5, 0xd62f105du) MD5_G(d, a, b, c, 10, 9, 0x02441453u) MD5_G(c, d, a, b, 15, 14, 0xd8a1e681u) MD5_G(b, c, d, a, 4, 20, 0xe7d3fbc8u)
	MD5_G(a, b, c, d, 9, 5, 0x21e1cde6u) MD5_G(d, a, b, c, 14, 9, 0xc33707d6u) MD5_G(c, d, a, b, 3, 14, 0xf4d50d87u) MD5_G(b, c, d, a, 8, 20, 0x455a14edu)
	MD5_G(a, b, c, d, 13, 5, 0xa9e3e905u) MD5_G(d, a, b, c, 2, 9, 0xfcefa3f8u) MD5_G(c, d, a, b, 7, 14, 0x676f02d9u) MD5_G(b, c, d, a, 12, 20, 0x8d2a4c8au)
	MD5_H(a, b, c, d, 5, 4, 0xfffa3942u) MD5_H(d, a, b, c, 8, 11, 0x8771f681u) MD5_H(c, d, a, b, 11, 16, 0x6d9d6122u) MD5_H(b, c, d, a, 14, 23, 0xfde5380cu)
	MD5_H(a, b, c, d, 1, 4, 0xa4beea44u) MD5_H(d, a, b, c, 4, 11, 0x4bdecfa9u) MD5_H(c, d, a, b, 7, 16, 0xf6bb4b60u) MD5_H(b, c, d, a, 10, 23, 0xbebfbc70u)
	MD5_H(a, b, c, d, 13, 4, 0x289b7ec6u) MD5_H(d, a, b, c, 0, 11, 0xeaa127fau) MD5_H(c, d, a, b, 3, 16, 0xd4ef3085u) MD5_H(b, c, d, a, 6, 23, 0x04881d05u)
	MD5_H(a, b, c, d, 9, 4, 0xd9d4d039u) MD5_H(d, a, b, c, 12, 11, 0xe6db99e5u) MD5_H(c, d, a, b, 15, 16, 0x1fa27cf8u) MD5_H(b, c, d, a, 2, 23, 0xc4ac5665u)
	MD5_I(a, b, c, d, 0, 6, 0xf4292244u) MD5_I(d, a, b, c, 7, 10, 0x432aff97u) MD5_I(c, d, a, b, 14, 15, 0xab9423a7u) MD5_I(b, c, d, a, 5, 21, 0xfc93a039u)
	MD5_I(a, b, c, d, 12, 6, 0x655b59c3u) MD5_I(d, a, b, c, 3, 10, 0x8f0ccc92u) MD5_I(c, d, a, b, 10, 15, 0xffeff47du) MD5_I(b, c, d, a, 1, 21, 0x85845dd1u)
	MD5_I(a, b, c, d, 8, 6, 0x6fa87e4fu) MD5_I(d, a, b, c, 15, 10, 0xfe2ce6e0u) MD5_I(c, d, a, b, 6, 15, 0xa3014314u) MD5_I(b, c, d, a, 13, 21, 0x4e0811a1u)
	MD5_I(a, b, c, d, 4, 6, 0xf7537e82u) MD5_I(d, a, b, c, 11, 10, 0xbd3af235u) MD5_I(c, d, a, b, 2, 15, 0x2ad7d2bbu) MD5_I(b, c, d, a, 9, 21, 0xeb86d391u)
#undef MD5_F
#undef MD5_G
#undef MD5_H
#undef MD5_I
	h_[0] += a;
	h_[1] += b;
	h_[2] += c;
	h_[3] += d;
}

} // namespace lrzgpu
