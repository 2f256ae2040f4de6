/* lz4_oracle.c -- ORACLE (test infrastructure; see lrzo.h).
 *
 * lz4_compresses(): reference src/stream.c:2325-2380.
 * LZ4_compress_default(): liblz4 is a system dependency absent from /root/reference
 * ("install liblz4-dev", configure.ac:139-140; unpinned). The container ships
 * liblz4 1.9.3; this file restates the published algorithm of that version
 * (lz4.c: LZ4_compress_fast_extState -> LZ4_compress_generic_validated, acceleration 1,
 * limitedOutput, noDict, byU16 below 64 KB+11 else byU32 with the 5-byte hash on
 * 64-bit little-endian hosts) as a SIZE-ONLY computation: the gate only needs the
 * return value.  tests/ cross-check it against the container's liblz4.so.1.
 */
#include <stdlib.h>
#include <string.h>
#include "lrzo.h"

#define MINMATCH 4
#define LASTLITERALS 5
#define MFLIMIT 12
#define LZ4_MINLENGTH (MFLIMIT + 1)
#define LZ4_64KLIMIT (65536 + (MFLIMIT - 1))
#define LZ4_SKIPTRIGGER 6
#define LZ4_HASHLOG 12
#define ML_MASK 15u
#define RUN_MASK 15u
#define LZ4_DISTANCE_MAX 65535
#define LZ4_MAX_INPUT_SIZE 0x7E000000

static inline uint32_t rd32(const uchar *p) { uint32_t v; memcpy(&v, p, 4); return v; }
static inline uint64_t rd64(const uchar *p) { uint64_t v; memcpy(&v, p, 8); return v; }

static inline uint32_t hash_pos(const uchar *p, int by_u16)
{
	if (by_u16)
		return (rd32(p) * 2654435761U) >> (MINMATCH * 8 - (LZ4_HASHLOG + 1));
	return (uint32_t)(((rd64(p) << 24) * 889523592379ULL) >> (64 - LZ4_HASHLOG));
}

static inline unsigned count_eq(const uchar *a, const uchar *b, const uchar *alimit)
{
	const uchar *s = a;
	while (a < alimit && *a == *b) {
		a++;
		b++;
	}
	return (unsigned)(a - s);
}

static int lz4_size_impl(const uchar *src, int src_size, int dst_capacity, int stop_below, int *stopped_early)
{
	/* all positions are offsets from src; op is the output byte count */
	uint32_t table[1 << (LZ4_HASHLOG + 1)];
	int by_u16, limited;
	i64 op = 0, olimit = dst_capacity;
	const uchar *ip = src, *anchor = src, *iend, *mflimit_plus_one, *matchlimit;
	uint32_t forward_h;

	if ((uint32_t)src_size > (uint32_t)LZ4_MAX_INPUT_SIZE)
		return 0;
	if (src_size == 0)
		return dst_capacity > 0 ? 1 : 0;
	{
		i64 bound = (i64)src_size + src_size / 255 + 16;
		limited = dst_capacity < bound;
	}
	by_u16 = src_size < LZ4_64KLIMIT;
	memset(table, 0, sizeof(table));
	iend = src + src_size;
	mflimit_plus_one = iend - MFLIMIT + 1;
	matchlimit = iend - LASTLITERALS;

	if (src_size < LZ4_MINLENGTH)
		goto last_literals;

	table[hash_pos(ip, by_u16)] = 0;
	ip++;
	forward_h = hash_pos(ip, by_u16);

	for (;;) {
		const uchar *match;
		/* find a match */
		{
			const uchar *forward_ip = ip;
			int step = 1;
			int search_nb = 1 << LZ4_SKIPTRIGGER;
			for (;;) {
				uint32_t h = forward_h;
				uint32_t current = (uint32_t)(forward_ip - src);
				uint32_t match_index = table[h];
				ip = forward_ip;
				forward_ip += step;
				step = search_nb++ >> LZ4_SKIPTRIGGER;
				if (forward_ip > mflimit_plus_one)
					goto last_literals;
				match = src + match_index;
				forward_h = hash_pos(forward_ip, by_u16);
				table[h] = current;
				if (!by_u16 && match_index + LZ4_DISTANCE_MAX < current)
					continue;
				if (rd32(match) == rd32(ip))
					break;
			}
		}
		/* catch up */
		while (ip > anchor && match > src && ip[-1] == match[-1]) {
			ip--;
			match--;
		}
		/* literals */
		{
			unsigned lit = (unsigned)(ip - anchor);
			op++; /* token */
			if (limited && op + lit + (2 + 1 + LASTLITERALS) + (lit / 255) > olimit)
				return 0;
			if (lit >= RUN_MASK)
				op += (lit - RUN_MASK) / 255 + 1;
			op += lit;
		}
	next_match:
		op += 2; /* offset */
		{
			unsigned mc = count_eq(ip + MINMATCH, match + MINMATCH, matchlimit);
			ip += (size_t)mc + MINMATCH;
			if (limited && op + (1 + LASTLITERALS) + (mc + 240) / 255 > olimit)
				return 0;
			if (mc >= ML_MASK)
				op += (mc - ML_MASK) / 255 + 1;
		}
		anchor = ip;
		if (stop_below > 0) {
			/* what is left can cost at most rest + rest/255 + 16 bytes (the LZ4_compressBound
			 * argument applied to the suffix): once that is below the caller's bound the verdict
			 * "smaller than stop_below" is certain */
			i64 rest = (i64)(iend - anchor);
			i64 ub = op + rest + rest / 255 + 16;
			if (ub < (i64)stop_below) {
				if (stopped_early)
					*stopped_early = 1;
				return (int)ub;
			}
		}
		if (ip >= mflimit_plus_one)
			break;
		table[hash_pos(ip - 2, by_u16)] = (uint32_t)(ip - 2 - src);
		{
			uint32_t h = hash_pos(ip, by_u16);
			uint32_t current = (uint32_t)(ip - src);
			uint32_t match_index = table[h];
			match = src + match_index;
			table[h] = current;
			if ((by_u16 || match_index + LZ4_DISTANCE_MAX >= current) && rd32(match) == rd32(ip)) {
				op++; /* token, zero literals */
				goto next_match;
			}
		}
		forward_h = hash_pos(++ip, by_u16);
	}

last_literals:
	{
		i64 last_run = (i64)(iend - anchor);
		if (limited && op + last_run + 1 + ((last_run + 255 - RUN_MASK) / 255) > olimit)
			return 0;
		if (last_run >= (i64)RUN_MASK)
			op += 1 + (last_run - RUN_MASK) / 255 + 1;
		else
			op += 1;
		op += last_run;
	}
	return (int)op;
}

int lrzo_lz4_compresses(const uchar *s_buf, i64 s_len, int threshold)
{
	const i64 ONE_MB = 1048576, STREAM_BUFSIZE = 10 * ONE_MB;
	i64 test_len = s_len;
	int in_len, d_len, buftest_size;
	double pct = 101;

	in_len = (int)(test_len < 100 * ONE_MB ? test_len : 100 * ONE_MB);
	buftest_size = in_len;
	d_len = in_len + 1;
	while (test_len > 0) {
		int r = lrzo_lz4_compress_default_size(s_buf, in_len, d_len);
		if (r > 0) {
			pct = 100 * ((double)r / (double)in_len);
			if (r < in_len * ((double)threshold / 100))
				break;
		}
		test_len -= in_len;
		if (test_len > 0) {
			buftest_size += in_len;
			if (buftest_size < STREAM_BUFSIZE)
				buftest_size <<= 1;
			in_len = (int)(test_len < buftest_size ? test_len : buftest_size);
			d_len = in_len + 1;
		}
	}
	return (int)(pct > threshold ? 0 : pct < 1 ? pct + 1 : pct);
}

int lrzo_lz4_compress_default_size(const uchar *src, int src_size, int dst_capacity)
{
	return lz4_size_impl(src, src_size, dst_capacity, 0, NULL);
}

/* The early-verdict variant the GPU gate uses in the pipeline (lz4_gate.hip, Lz4Job.stop_below):
 * returns the exact size, or -- as soon as "size < stop_below" is certain -- an upper bound of the
 * size that is itself below stop_below.  *stopped_early tells which. */
int lrzo_lz4_size_stop_below(const uchar *src, int src_size, int dst_capacity, int stop_below, int *stopped_early)
{
	if (stopped_early)
		*stopped_early = 0;
	return lz4_size_impl(src, src_size, dst_capacity, stop_below, stopped_early);
}
