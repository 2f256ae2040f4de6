/* hashes_oracle.c -- ORACLE (test infrastructure; see lrzo.h).
 * CRC-32/IEEE and MD5 (RFC 1321), the two digests the reference obtains from
 * libgcrypt (src/rzip.c:571-573, 741-761, 1195-1219).  libgcrypt is a system
 * library not under /root/reference; both algorithms are public standards. */
#include <string.h>
#include "lrzo.h"

static uint32_t crc_tab[8][256];
static int crc_ready;

static void crc_init(void)
{
	unsigned i, j;
	for (i = 0; i < 256; i++) {
		uint32_t r = i;
		for (j = 0; j < 8; j++)
			r = (r >> 1) ^ (0xEDB88320u & (0u - (r & 1)));
		crc_tab[0][i] = r;
	}
	for (i = 0; i < 256; i++)
		for (j = 1; j < 8; j++)
			crc_tab[j][i] = (crc_tab[j - 1][i] >> 8) ^ crc_tab[0][crc_tab[j - 1][i] & 0xFF];
	crc_ready = 1;
}

uint32_t lrzo_crc32(uint32_t crc, const uchar *p, size_t n)
{
	if (!crc_ready)
		crc_init();
	crc = ~crc;
	while (n >= 8) {
		uint32_t a, b;
		memcpy(&a, p, 4);
		memcpy(&b, p + 4, 4);
		a ^= crc;
		crc = crc_tab[7][a & 0xFF] ^ crc_tab[6][(a >> 8) & 0xFF] ^ crc_tab[5][(a >> 16) & 0xFF] ^
		      crc_tab[4][a >> 24] ^ crc_tab[3][b & 0xFF] ^ crc_tab[2][(b >> 8) & 0xFF] ^
		      crc_tab[1][(b >> 16) & 0xFF] ^ crc_tab[0][b >> 24];
		p += 8;
		n -= 8;
	}
	while (n--)
		crc = (crc >> 8) ^ crc_tab[0][(crc ^ *p++) & 0xFF];
	return ~crc;
}

static const uint32_t K[64] = {
	0xd76aa478, 0xe8c7b756, 0x242070db, 0xc1bdceee, 0xf57c0faf, 0x4787c62a, 0xa8304613, 0xfd469501,
	0x698098d8, 0x8b44f7af, 0xffff5bb1, 0x895cd7be, 0x6b901122, 0xfd987193, 0xa679438e, 0x49b40821,
	0xf61e2562, 0xc040b340, 0x265e5a51, 0xe9b6c7aa, 0xd62f105d, 0x02441453, 0xd8a1e681, 0xe7d3fbc8,
	0x21e1cde6, 0xc33707d6, 0xf4d50d87, 0x455a14ed, 0xa9e3e905, 0xfcefa3f8, 0x676f02d9, 0x8d2a4c8a,
	0xfffa3942, 0x8771f681, 0x6d9d6122, 0xfde5380c, 0xa4beea44, 0x4bdecfa9, 0xf6bb4b60, 0xbebfbc70,
	0x289b7ec6, 0xeaa127fa, 0xd4ef3085, 0x04881d05, 0xd9d4d039, 0xe6db99e5, 0x1fa27cf8, 0xc4ac5665,
	0xf4292244, 0x432aff97, 0xab9423a7, 0xfc93a039, 0x655b59c3, 0x8f0ccc92, 0xffeff47d, 0x85845dd1,
	0x6fa87e4f, 0xfe2ce6e0, 0xa3014314, 0x4e0811a1, 0xf7537e82, 0xbd3af235, 0x2ad7d2bb, 0xeb86d391,
};
static const uchar S[64] = {
	7, 12, 17, 22, 7, 12, 17, 22, 7, 12, 17, 22, 7, 12, 17, 22, 5, 9, 14, 20, 5, 9, 14, 20, 5, 9, 14, 20, 5, 9, 14, 20,
	4, 11, 16, 23, 4, 11, 16, 23, 4, 11, 16, 23, 4, 11, 16, 23, 6, 10, 15, 21, 6, 10, 15, 21, 6, 10, 15, 21, 6, 10, 15, 21,
};

static void md5_block(lrzo_md5 *m, const uchar *p)
{
	uint32_t w[16], a = m->a, b = m->b, c = m->c, d = m->d;
	int i;
	for (i = 0; i < 16; i++)
		w[i] = (uint32_t)p[4 * i] | ((uint32_t)p[4 * i + 1] << 8) | ((uint32_t)p[4 * i + 2] << 16) | ((uint32_t)p[4 * i + 3] << 24);
	for (i = 0; i < 64; i++) {
		uint32_t f, t;
		int g;
		if (i < 16) { f = (b & c) | (~b & d); g = i; }
		else if (i < 32) { f = (d & b) | (~d & c); g = (5 * i + 1) & 15; }
		else if (i < 48) { f = b ^ c ^ d; g = (3 * i + 5) & 15; }
		else { f = c ^ (b | ~d); g = (7 * i) & 15; }
		t = a + f + K[i] + w[g];
		a = d; d = c; c = b;
		b += (t << S[i]) | (t >> (32 - S[i]));
	}
	m->a += a; m->b += b; m->c += c; m->d += d;
}

void lrzo_md5_init(lrzo_md5 *m)
{
	m->a = 0x67452301; m->b = 0xefcdab89; m->c = 0x98badcfe; m->d = 0x10325476;
	m->len = 0;
	m->fill = 0;
}

void lrzo_md5_update(lrzo_md5 *m, const uchar *p, size_t n)
{
	m->len += n;
	if (m->fill) {
		size_t k = 64 - m->fill;
		if (k > n)
			k = n;
		memcpy(m->buf + m->fill, p, k);
		m->fill += (unsigned)k;
		p += k;
		n -= k;
		if (m->fill < 64)
			return;
		md5_block(m, m->buf);
		m->fill = 0;
	}
	while (n >= 64) {
		md5_block(m, p);
		p += 64;
		n -= 64;
	}
	if (n) {
		memcpy(m->buf, p, n);
		m->fill = (unsigned)n;
	}
}

void lrzo_md5_final(lrzo_md5 *m, uchar out[16])
{
	uint64_t bits = m->len * 8;
	uchar pad[72];
	size_t padn = (m->fill < 56) ? 56 - m->fill : 120 - m->fill;
	int i;
	memset(pad, 0, sizeof(pad));
	pad[0] = 0x80;
	for (i = 0; i < 8; i++)
		pad[padn + i] = (uchar)(bits >> (8 * i));
	lrzo_md5_update(m, pad, padn + 8);
	for (i = 0; i < 4; i++) {
		out[i] = (uchar)(m->a >> (8 * i));
		out[4 + i] = (uchar)(m->b >> (8 * i));
		out[8 + i] = (uchar)(m->c >> (8 * i));
		out[12 + i] = (uchar)(m->d >> (8 * i));
	}
}
