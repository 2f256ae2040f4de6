// hashes.cpp -- see hashes.h.  Plain host C++ written from the specifications; every digest is checked against
// Python's hashlib in tests/test_hashes_cpu.py.
#include "hashes.h"

#include <cstring>

#include "md5.h"

namespace lrzgpu {
namespace {

inline uint32_t rotl32(uint32_t x, int k) { return (x << k) | (x >> (32 - k)); }
inline uint32_t rotr32(uint32_t x, int k) { return (x >> k) | (x << (32 - k)); }
inline uint64_t rotr64(uint64_t x, int k) { return (x >> k) | (x << (64 - k)); }
inline uint64_t rotl64(uint64_t x, int k) { return k ? (x << k) | (x >> (64 - k)) : x; }

// ---- a block hash with Merkle-Damgard padding: BLOCK bytes per compression, length in bits appended
// (LEN_BYTES wide, big or little endian) -------------------------------------------------------------------
template <size_t BLOCK> struct BlockBuffer {
	uint8_t buf[BLOCK];
	size_t fill = 0;
	uint64_t total = 0;
	template <class F> void update(const uint8_t *p, size_t n, F &&compress)
	{
		total += n;
		if (fill) {
			size_t k = BLOCK - fill;
			if (k > n)
				k = n;
			memcpy(buf + fill, p, k);
			fill += k;
			p += k;
			n -= k;
			if (fill < BLOCK)
				return;
			compress(buf);
			fill = 0;
		}
		for (; n >= BLOCK; p += BLOCK, n -= BLOCK)
			compress(p);
		if (n) {
			memcpy(buf, p, n);
			fill = n;
		}
	}
	// 0x80, zeros, then the bit length in the last LEN_BYTES of a block
	template <class F> void pad(size_t len_bytes, bool big_endian, F &&compress)
	{
		const uint64_t bits = total * 8; // (inputs beyond 2^61 bytes are not a concern here)
		buf[fill++] = 0x80;
		if (fill > BLOCK - len_bytes) {
			memset(buf + fill, 0, BLOCK - fill);
			compress(buf);
			fill = 0;
		}
		memset(buf + fill, 0, BLOCK - fill);
		for (size_t i = 0; i < 8; i++)
			buf[big_endian ? BLOCK - 1 - i : BLOCK - len_bytes + i] = (uint8_t)(bits >> (8 * i));
		compress(buf);
		fill = 0;
	}
};

// ---- MD5 (md5.h) ------------------------------------------------------------------------------------------
struct Md5Hasher : Hasher {
	Md5 m;
	void update(const uint8_t *p, size_t n) override { m.update(p, n); }
	void finish(uint8_t *out) override { m.finish(out); }
};

// ---- CRC-32 (ISO 3309 / IEEE 802.3, reflected 0xEDB88320; libgcrypt returns the value most significant byte first) ---
struct Crc32Hasher : Hasher {
	uint32_t table[8][256];
	uint32_t c = 0xFFFFFFFFu;
	Crc32Hasher()
	{
		for (uint32_t i = 0; i < 256; i++) {
			uint32_t v = i;
			for (int k = 0; k < 8; k++)
				v = (v >> 1) ^ (0xEDB88320u & (0u - (v & 1)));
			table[0][i] = v;
		}
		for (int t = 1; t < 8; t++)
			for (uint32_t i = 0; i < 256; i++)
				table[t][i] = (table[t - 1][i] >> 8) ^ table[0][table[t - 1][i] & 0xFF];
	}
	void update(const uint8_t *p, size_t n) override
	{
		uint32_t v = c;
		for (; n >= 8; p += 8, n -= 8) { // slicing by eight
			uint32_t lo, hi;
			memcpy(&lo, p, 4);
			memcpy(&hi, p + 4, 4);
			lo ^= v;
			v = table[7][lo & 0xFF] ^ table[6][(lo >> 8) & 0xFF] ^ table[5][(lo >> 16) & 0xFF] ^ table[4][lo >> 24] ^ table[3][hi & 0xFF] ^
			    table[2][(hi >> 8) & 0xFF] ^ table[1][(hi >> 16) & 0xFF] ^ table[0][hi >> 24];
		}
		for (; n; p++, n--)
			v = (v >> 8) ^ table[0][(v ^ *p) & 0xFF];
		c = v;
	}
	void finish(uint8_t *out) override
	{
		const uint32_t v = c ^ 0xFFFFFFFFu;
		for (int i = 0; i < 4; i++)
			out[i] = (uint8_t)(v >> (24 - 8 * i));
	}
};

// ---- RIPEMD-160 (Dobbertin, Bosselaers, Preneel 1996) ---------------------------------------------------------
struct Ripemd160Hasher : Hasher {
	uint32_t h[5] = {0x67452301u, 0xEFCDAB89u, 0x98BADCFEu, 0x10325476u, 0xC3D2E1F0u};
	BlockBuffer<64> bb;
	static uint32_t f(int j, uint32_t x, uint32_t y, uint32_t z)
	{
		switch (j >> 4) {
		case 0: return x ^ y ^ z;
		case 1: return (x & y) | (~x & z);
		case 2: return (x | ~y) ^ z;
		case 3: return (x & z) | (y & ~z);
		default: return x ^ (y | ~z);
		}
	}
	void compress(const uint8_t *p)
	{
		static const uint8_t RL[80] = {0, 1, 2,  3,  4,  5,  6,  7, 8,  9,  10, 11, 12, 13, 14, 15, 7, 4, 13, 1,  10, 6,  15, 3, 12, 0, 9,
					       5, 2, 14, 11, 8,  3,  10, 14, 4, 9,  15, 8,  1,  2,  7,  0,  6, 13, 11, 5, 12, 1,  9,  11, 10, 0, 8,
					       12, 4, 13, 3, 7,  15, 14, 5,  6, 2,  4,  0,  5,  9,  7,  12, 2, 10, 14, 1, 3,  8,  11, 6,  15, 13};
		static const uint8_t RR[80] = {5, 14, 7, 0, 9, 2,  11, 4,  13, 6,  15, 8, 1,  10, 3,  12, 6,  11, 3,  7,  0, 13, 5,  10, 14, 15, 8,
					       12, 4, 9, 1, 2, 15, 5,  1,  3,  7,  14, 6, 9,  11, 8,  12, 2,  10, 0,  4,  13, 8, 6,  4,  1,  3,  11, 15,
					       0, 5, 12, 2, 13, 9, 7,  10, 14, 12, 15, 10, 4, 1,  5,  8,  7,  6,  2,  13, 14, 0, 3,  9,  11};
		static const uint8_t SL[80] = {11, 14, 15, 12, 5,  8,  7,  9,  11, 13, 14, 15, 6,  7,  9,  8,  7,  6,  8,  13, 11, 9,  7,  15, 7,  12, 15,
					       9,  11, 7,  13, 12, 11, 13, 6,  7,  14, 9,  13, 15, 14, 8,  13, 6,  5,  12, 7,  5,  11, 12, 14, 15, 14, 15,
					       9,  8,  9,  14, 5,  6,  8,  6,  5,  12, 9,  15, 5,  11, 6,  8,  13, 12, 5,  12, 13, 14, 11, 8,  5,  6};
		static const uint8_t SR[80] = {8,  9,  9,  11, 13, 15, 15, 5,  7,  7,  8,  11, 14, 14, 12, 6,  9,  13, 15, 7,  12, 8,  9,  11, 7,  7,  12,
					       7,  6,  15, 13, 11, 9,  7,  15, 11, 8,  6,  6,  14, 12, 13, 5,  14, 13, 13, 7,  5,  15, 5,  8,  11, 14, 14,
					       6,  14, 6,  9,  12, 9,  12, 5,  15, 8,  8,  5,  12, 9,  12, 5,  14, 6,  8,  13, 6,  5,  15, 13, 11, 11};
		static const uint32_t KL[5] = {0x00000000u, 0x5A827999u, 0x6ED9EBA1u, 0x8F1BBCDCu, 0xA953FD4Eu};
		static const uint32_t KR[5] = {0x50A28BE6u, 0x5C4DD124u, 0x6D703EF3u, 0x7A6D76E9u, 0x00000000u};
		uint32_t x[16];
		for (int i = 0; i < 16; i++)
			x[i] = (uint32_t)p[4 * i] | (uint32_t)p[4 * i + 1] << 8 | (uint32_t)p[4 * i + 2] << 16 | (uint32_t)p[4 * i + 3] << 24;
		uint32_t al = h[0], bl = h[1], cl = h[2], dl = h[3], el = h[4];
		uint32_t ar = al, br = bl, cr = cl, dr = dl, er = el;
		for (int j = 0; j < 80; j++) {
			uint32_t t = rotl32(al + f(j, bl, cl, dl) + x[RL[j]] + KL[j >> 4], SL[j]) + el;
			al = el;
			el = dl;
			dl = rotl32(cl, 10);
			cl = bl;
			bl = t;
			t = rotl32(ar + f(79 - j, br, cr, dr) + x[RR[j]] + KR[j >> 4], SR[j]) + er;
			ar = er;
			er = dr;
			dr = rotl32(cr, 10);
			cr = br;
			br = t;
		}
		const uint32_t t = h[1] + cl + dr;
		h[1] = h[2] + dl + er;
		h[2] = h[3] + el + ar;
		h[3] = h[4] + al + br;
		h[4] = h[0] + bl + cr;
		h[0] = t;
	}
	void update(const uint8_t *p, size_t n) override
	{
		bb.update(p, n, [this](const uint8_t *b) { compress(b); });
	}
	void finish(uint8_t *out) override
	{
		bb.pad(8, false, [this](const uint8_t *b) { compress(b); });
		for (int w = 0; w < 5; w++)
			for (int i = 0; i < 4; i++)
				out[4 * w + i] = (uint8_t)(h[w] >> (8 * i));
	}
};

// ---- SHA-256 (FIPS 180-4) ---------------------------------------------------------------------------------------
struct Sha256Hasher : Hasher {
	uint32_t h[8] = {0x6a09e667u, 0xbb67ae85u, 0x3c6ef372u, 0xa54ff53au, 0x510e527fu, 0x9b05688cu, 0x1f83d9abu, 0x5be0cd19u};
	BlockBuffer<64> bb;
	void compress(const uint8_t *p)
	{
		static const uint32_t K[64] = {
			0x428a2f98u, 0x71374491u, 0xb5c0fbcfu, 0xe9b5dba5u, 0x3956c25bu, 0x59f111f1u, 0x923f82a4u, 0xab1c5ed5u, 0xd807aa98u, 0x12835b01u, 0x243185beu,
			0x550c7dc3u, 0x72be5d74u, 0x80deb1feu, 0x9bdc06a7u, 0xc19bf174u, 0xe49b69c1u, 0xefbe4786u, 0x0fc19dc6u, 0x240ca1ccu, 0x2de92c6fu, 0x4a7484aau,
			0x5cb0a9dcu, 0x76f988dau, 0x983e5152u, 0xa831c66du, 0xb00327c8u, 0xbf597fc7u, 0xc6e00bf3u, 0xd5a79147u, 0x06ca6351u, 0x14292967u, 0x27b70a85u,
			0x2e1b2138u, 0x4d2c6dfcu, 0x53380d13u, 0x650a7354u, 0x766a0abbu, 0x81c2c92eu, 0x92722c85u, 0xa2bfe8a1u, 0xa81a664bu, 0xc24b8b70u, 0xc76c51a3u,
			0xd192e819u, 0xd6990624u, 0xf40e3585u, 0x106aa070u, 0x19a4c116u, 0x1e376c08u, 0x2748774cu, 0x34b0bcb5u, 0x391c0cb3u, 0x4ed8aa4au, 0x5b9cca4fu,
			0x682e6ff3u, 0x748f82eeu, 0x78a5636fu, 0x84c87814u, 0x8cc70208u, 0x90befffau, 0xa4506cebu, 0xbef9a3f7u, 0xc67178f2u};
		uint32_t w[64];
		for (int i = 0; i < 16; i++)
			w[i] = (uint32_t)p[4 * i] << 24 | (uint32_t)p[4 * i + 1] << 16 | (uint32_t)p[4 * i + 2] << 8 | (uint32_t)p[4 * i + 3];
		for (int i = 16; i < 64; i++) {
			const uint32_t s0 = rotr32(w[i - 15], 7) ^ rotr32(w[i - 15], 18) ^ (w[i - 15] >> 3);
			const uint32_t s1 = rotr32(w[i - 2], 17) ^ rotr32(w[i - 2], 19) ^ (w[i - 2] >> 10);
			w[i] = w[i - 16] + s0 + w[i - 7] + s1;
		}
		uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
		for (int i = 0; i < 64; i++) {
			const uint32_t t1 = hh + (rotr32(e, 6) ^ rotr32(e, 11) ^ rotr32(e, 25)) + ((e & f) ^ (~e & g)) + K[i] + w[i];
			const uint32_t t2 = (rotr32(a, 2) ^ rotr32(a, 13) ^ rotr32(a, 22)) + ((a & b) ^ (a & c) ^ (b & c));
			hh = g;
			g = f;
			f = e;
			e = d + t1;
			d = c;
			c = b;
			b = a;
			a = t1 + t2;
		}
		h[0] += a;
		h[1] += b;
		h[2] += c;
		h[3] += d;
		h[4] += e;
		h[5] += f;
		h[6] += g;
		h[7] += hh;
	}
	void update(const uint8_t *p, size_t n) override
	{
		bb.update(p, n, [this](const uint8_t *b) { compress(b); });
	}
	void finish(uint8_t *out) override
	{
		bb.pad(8, true, [this](const uint8_t *b) { compress(b); });
		for (int w = 0; w < 8; w++)
			for (int i = 0; i < 4; i++)
				out[4 * w + i] = (uint8_t)(h[w] >> (24 - 8 * i));
	}
};

// ---- SHA-512 / SHA-384 (FIPS 180-4) --------------------------------------------------------------------------------
struct Sha512Hasher : Hasher {
	uint64_t h[8];
	size_t out_len;
	BlockBuffer<128> bb;
	explicit Sha512Hasher(bool sha384) : out_len(sha384 ? 48 : 64)
	{
		static const uint64_t I512[8] = {0x6a09e667f3bcc908ull, 0xbb67ae8584caa73bull, 0x3c6ef372fe94f82bull, 0xa54ff53a5f1d36f1ull,
						 0x510e527fade682d1ull, 0x9b05688c2b3e6c1full, 0x1f83d9abfb41bd6bull, 0x5be0cd19137e2179ull};
		static const uint64_t I384[8] = {0xcbbb9d5dc1059ed8ull, 0x629a292a367cd507ull, 0x9159015a3070dd17ull, 0x152fecd8f70e5939ull,
						 0x67332667ffc00b31ull, 0x8eb44a8768581511ull, 0xdb0c2e0d64f98fa7ull, 0x47b5481dbefa4fa4ull};
		memcpy(h, sha384 ? I384 : I512, sizeof(h));
	}
	void compress(const uint8_t *p)
	{
		static const uint64_t K[80] = {
			0x428a2f98d728ae22ull, 0x7137449123ef65cdull, 0xb5c0fbcfec4d3b2full, 0xe9b5dba58189dbbcull, 0x3956c25bf348b538ull, 0x59f111f1b605d019ull,
			0x923f82a4af194f9bull, 0xab1c5ed5da6d8118ull, 0xd807aa98a3030242ull, 0x12835b0145706fbeull, 0x243185be4ee4b28cull, 0x550c7dc3d5ffb4e2ull,
			0x72be5d74f27b896full, 0x80deb1fe3b1696b1ull, 0x9bdc06a725c71235ull, 0xc19bf174cf692694ull, 0xe49b69c19ef14ad2ull, 0xefbe4786384f25e3ull,
			0x0fc19dc68b8cd5b5ull, 0x240ca1cc77ac9c65ull, 0x2de92c6f592b0275ull, 0x4a7484aa6ea6e483ull, 0x5cb0a9dcbd41fbd4ull, 0x76f988da831153b5ull,
			0x983e5152ee66dfabull, 0xa831c66d2db43210ull, 0xb00327c898fb213full, 0xbf597fc7beef0ee4ull, 0xc6e00bf33da88fc2ull, 0xd5a79147930aa725ull,
			0x06ca6351e003826full, 0x142929670a0e6e70ull, 0x27b70a8546d22ffcull, 0x2e1b21385c26c926ull, 0x4d2c6dfc5ac42aedull, 0x53380d139d95b3dfull,
			0x650a73548baf63deull, 0x766a0abb3c77b2a8ull, 0x81c2c92e47edaee6ull, 0x92722c851482353bull, 0xa2bfe8a14cf10364ull, 0xa81a664bbc423001ull,
			0xc24b8b70d0f89791ull, 0xc76c51a30654be30ull, 0xd192e819d6ef5218ull, 0xd69906245565a910ull, 0xf40e35855771202aull, 0x106aa07032bbd1b8ull,
			0x19a4c116b8d2d0c8ull, 0x1e376c085141ab53ull, 0x2748774cdf8eeb99ull, 0x34b0bcb5e19b48a8ull, 0x391c0cb3c5c95a63ull, 0x4ed8aa4ae3418acbull,
			0x5b9cca4f7763e373ull, 0x682e6ff3d6b2b8a3ull, 0x748f82ee5defb2fcull, 0x78a5636f43172f60ull, 0x84c87814a1f0ab72ull, 0x8cc702081a6439ecull,
			0x90befffa23631e28ull, 0xa4506cebde82bde9ull, 0xbef9a3f7b2c67915ull, 0xc67178f2e372532bull, 0xca273eceea26619cull, 0xd186b8c721c0c207ull,
			0xeada7dd6cde0eb1eull, 0xf57d4f7fee6ed178ull, 0x06f067aa72176fbaull, 0x0a637dc5a2c898a6ull, 0x113f9804bef90daeull, 0x1b710b35131c471bull,
			0x28db77f523047d84ull, 0x32caab7b40c72493ull, 0x3c9ebe0a15c9bebcull, 0x431d67c49c100d4cull, 0x4cc5d4becb3e42b6ull, 0x597f299cfc657e2aull,
			0x5fcb6fab3ad6faecull, 0x6c44198c4a475817ull};
		uint64_t w[80];
		for (int i = 0; i < 16; i++) {
			uint64_t v = 0;
			for (int k = 0; k < 8; k++)
				v = v << 8 | p[8 * i + k];
			w[i] = v;
		}
		for (int i = 16; i < 80; i++) {
			const uint64_t s0 = rotr64(w[i - 15], 1) ^ rotr64(w[i - 15], 8) ^ (w[i - 15] >> 7);
			const uint64_t s1 = rotr64(w[i - 2], 19) ^ rotr64(w[i - 2], 61) ^ (w[i - 2] >> 6);
			w[i] = w[i - 16] + s0 + w[i - 7] + s1;
		}
		uint64_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
		for (int i = 0; i < 80; i++) {
			const uint64_t t1 = hh + (rotr64(e, 14) ^ rotr64(e, 18) ^ rotr64(e, 41)) + ((e & f) ^ (~e & g)) + K[i] + w[i];
			const uint64_t t2 = (rotr64(a, 28) ^ rotr64(a, 34) ^ rotr64(a, 39)) + ((a & b) ^ (a & c) ^ (b & c));
			hh = g;
			g = f;
			f = e;
			e = d + t1;
			d = c;
			c = b;
			b = a;
			a = t1 + t2;
		}
		h[0] += a;
		h[1] += b;
		h[2] += c;
		h[3] += d;
		h[4] += e;
		h[5] += f;
		h[6] += g;
		h[7] += hh;
	}
	void update(const uint8_t *p, size_t n) override
	{
		bb.update(p, n, [this](const uint8_t *b) { compress(b); });
	}
	void finish(uint8_t *out) override
	{
		bb.pad(16, true, [this](const uint8_t *b) { compress(b); }); // (the high 64 bits of the 128-bit length stay zero)
		uint8_t full[64];
		for (int w = 0; w < 8; w++)
			for (int i = 0; i < 8; i++)
				full[8 * w + i] = (uint8_t)(h[w] >> (56 - 8 * i));
		memcpy(out, full, out_len);
	}
};

// ---- Keccak-f[1600] sponge: SHA3-256/512 (suffix 0x06) and SHAKE128/256 (suffix 0x1F), FIPS 202 ------------------------
struct KeccakHasher : Hasher {
	uint64_t a[25];
	uint8_t buf[168];
	size_t rate, fill = 0, out_len;
	uint8_t suffix;
	KeccakHasher(size_t rate_bytes, uint8_t dsuffix, size_t out_bytes) : rate(rate_bytes), out_len(out_bytes), suffix(dsuffix) { memset(a, 0, sizeof(a)); }
	void permute()
	{
		static const uint64_t RC[24] = {0x0000000000000001ull, 0x0000000000008082ull, 0x800000000000808aull, 0x8000000080008000ull, 0x000000000000808bull,
						0x0000000080000001ull, 0x8000000080008081ull, 0x8000000000008009ull, 0x000000000000008aull, 0x0000000000000088ull,
						0x0000000080008009ull, 0x000000008000000aull, 0x000000008000808bull, 0x800000000000008bull, 0x8000000000008089ull,
						0x8000000000008003ull, 0x8000000000008002ull, 0x8000000000000080ull, 0x000000000000800aull, 0x800000008000000aull,
						0x8000000080008081ull, 0x8000000000008080ull, 0x0000000080000001ull, 0x8000000080008008ull};
		static const int ROT[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14}; // [x + 5 y]
		for (int round = 0; round < 24; round++) {
			uint64_t c[5], d[5], b[25];
			for (int x = 0; x < 5; x++)
				c[x] = a[x] ^ a[x + 5] ^ a[x + 10] ^ a[x + 15] ^ a[x + 20];
			for (int x = 0; x < 5; x++)
				d[x] = c[(x + 4) % 5] ^ rotl64(c[(x + 1) % 5], 1);
			for (int i = 0; i < 25; i++)
				a[i] ^= d[i % 5];
			// rho + pi: B[y, 2x + 3y] = rot(A[x, y])
			for (int x = 0; x < 5; x++)
				for (int y = 0; y < 5; y++)
					b[y + 5 * ((2 * x + 3 * y) % 5)] = rotl64(a[x + 5 * y], ROT[x + 5 * y]);
			for (int y = 0; y < 5; y++)
				for (int x = 0; x < 5; x++)
					a[x + 5 * y] = b[x + 5 * y] ^ (~b[(x + 1) % 5 + 5 * y] & b[(x + 2) % 5 + 5 * y]);
			a[0] ^= RC[round];
		}
	}
	void absorb_block(const uint8_t *p)
	{
		for (size_t i = 0; i < rate / 8; i++) {
			uint64_t v;
			memcpy(&v, p + 8 * i, 8); // little-endian lanes (x86)
			a[i] ^= v;
		}
		permute();
	}
	void update(const uint8_t *p, size_t n) override
	{
		if (fill) {
			size_t k = rate - fill;
			if (k > n)
				k = n;
			memcpy(buf + fill, p, k);
			fill += k;
			p += k;
			n -= k;
			if (fill < rate)
				return;
			absorb_block(buf);
			fill = 0;
		}
		for (; n >= rate; p += rate, n -= rate)
			absorb_block(p);
		if (n) {
			memcpy(buf, p, n);
			fill = n;
		}
	}
	void finish(uint8_t *out) override
	{
		memset(buf + fill, 0, rate - fill);
		buf[fill] ^= suffix;
		buf[rate - 1] ^= 0x80;
		absorb_block(buf);
		size_t done = 0;
		while (done < out_len) { // squeeze (every output here fits the first block; kept general)
			size_t k = out_len - done < rate ? out_len - done : rate;
			memcpy(out + done, a, k);
			done += k;
			if (done < out_len)
				permute();
		}
	}
};

const int kLen[HASH_MAX + 1] = {4, 16, 20, 32, 48, 64, 32, 64, 16, 32, 64, 16, 32, 64};
const char *const kLabel[HASH_MAX + 1] = {"CRC",      "MD5",	 "RIPEMD",	"SHA256",      "SHA384",	"SHA512",      "SHA3_256",
					  "SHA3_512", "SHAKE128_16", "SHAKE128_32", "SHAKE128_64", "SHAKE256_16", "SHAKE256_32", "SHAKE256_64"};

} // namespace

int hash_length(int code) { return code < 0 || code > HASH_MAX ? -1 : kLen[code]; }
const char *hash_label(int code) { return code < 0 || code > HASH_MAX ? "?" : kLabel[code]; }

std::unique_ptr<Hasher> make_hasher(int code)
{
	switch (code) {
	case HASH_CRC: return std::unique_ptr<Hasher>(new Crc32Hasher());
	case HASH_MD5: return std::unique_ptr<Hasher>(new Md5Hasher());
	case HASH_RIPEMD: return std::unique_ptr<Hasher>(new Ripemd160Hasher());
	case HASH_SHA256: return std::unique_ptr<Hasher>(new Sha256Hasher());
	case HASH_SHA384: return std::unique_ptr<Hasher>(new Sha512Hasher(true));
	case HASH_SHA512: return std::unique_ptr<Hasher>(new Sha512Hasher(false));
	case HASH_SHA3_256: return std::unique_ptr<Hasher>(new KeccakHasher(136, 0x06, 32));
	case HASH_SHA3_512: return std::unique_ptr<Hasher>(new KeccakHasher(72, 0x06, 64));
	case HASH_SHAKE128_16:
	case HASH_SHAKE128_32:
	case HASH_SHAKE128_64: return std::unique_ptr<Hasher>(new KeccakHasher(168, 0x1F, (size_t)kLen[code]));
	case HASH_SHAKE256_16:
	case HASH_SHAKE256_32:
	case HASH_SHAKE256_64: return std::unique_ptr<Hasher>(new KeccakHasher(136, 0x1F, (size_t)kLen[code]));
	default: return nullptr;
	}
}

} // namespace lrzgpu
