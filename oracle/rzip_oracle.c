/* rzip_oracle.c -- ORACLE (test infrastructure; see lrzo.h).
 *
 * CPU restatement of lrzip-next's rzip scan: reference src/rzip.c
 *   hash_search            586-762    -> lrzo_rzip_chunk
 *   insert_hash            304-353    -> tbl_insert
 *   clean_one_from_hash    357-383    -> tbl_clean_one
 *   find_best_match        495-534    -> tbl_lookup
 *   single_match_len       431-461    -> match_extent
 *   single_full_tag / next 385-416    -> window_tag / rolling update in the main loop
 *   put_match / put_literal 208-265   -> emit_match / emit_literal
 */
#include <stdlib.h>
#include <string.h>
#include "lrzo.h"

#define MINIMUM_MATCH 31   /* src/rzip.c:51 */
#define GREAT_MATCH 1024   /* src/rzip.c:50 */

static const lrzo_level LEVELS[10] = { /* src/rzip.c:67-82 */
	{1, 4, 1}, {2, 4, 2}, {4, 4, 2}, {8, 4, 2}, {16, 4, 3},
	{32, 4, 4}, {32, 2, 6}, {64, 1, 16}, {64, 1, 32}, {64, 1, 128},
};

const lrzo_level *lrzo_rzip_level(int level)
{
	if (level < 0 || level > 9)
		return NULL;
	return &LEVELS[level];
}

void lrzo_hash_index(uint64_t out[256])
{
	/* glibc random() TYPE_3, default seed 1 (the reference never calls
	 * srandom). r[i] = r[i-3] + r[i-31], output >> 1, after discarding 310.
	 * Restated here so the table does not depend on the process' random() state. */
	int32_t r[34 + 310 + 512];
	int i, k = 0;
	r[0] = 1;
	for (i = 1; i < 31; i++) {
		/* 16807 * r[i-1] % 2147483647 via Schrage, as glibc srandom_r */
		long hi = r[i - 1] / 127773, lo = r[i - 1] % 127773;
		long w = 16807 * lo - 2836 * hi;
		if (w < 0)
			w += 2147483647;
		r[i] = (int32_t)w;
	}
	for (i = 31; i < 34; i++)
		r[i] = r[i - 31];
	for (i = 34; i < 34 + 310 + 512; i++)
		r[i] = (int32_t)((uint32_t)r[i - 31] + (uint32_t)r[i - 3]);
	for (i = 0; i < 256; i++) {
		uint64_t a = ((uint32_t)r[34 + 310 + k++]) >> 1;
		uint64_t b = ((uint32_t)r[34 + 310 + k++]) >> 1;
		out[i] = (a << 16) ^ b;
	}
}

struct slot { i64 offset; uint64_t t; };

struct scan {
	const uchar *buf;
	i64 chunk_size;
	int chunk_bytes;
	const uint64_t *hx;
	const lrzo_level *lvl;
	struct slot *tbl;
	int hash_bits;
	i64 hash_limit, hash_count;
	uint64_t minimum_tag_mask;
	i64 tag_clean_ptr;
	i64 last_match;
	i64 *victim_round;
	const lrzo_sink *sink;
	lrzo_rzip_stats st;
};

static inline int slot_empty(const struct slot *s) { return !(s->offset | (i64)s->t); }
static inline uint64_t mask_up(uint64_t m) { return (m << 1) | 1; }
static inline i64 bucket_of(const struct scan *s, uint64_t t) { return (i64)(t & (((uint64_t)1 << s->hash_bits) - 1)); }

/* ffsll(~t): 1 + number of trailing one bits (0 when t is all ones) */
static inline int bitness_rank(uint64_t t)
{
	uint64_t v = ~t;
	return v ? __builtin_ctzll(v) + 1 : 0;
}

static inline int below_next_mask(const struct scan *s, uint64_t t)
{
	uint64_t better = mask_up(s->minimum_tag_mask);
	return (t & better) != better;
}

static void tbl_insert(struct scan *s, uint64_t t, i64 offset)
{
	i64 h = bucket_of(s, t), victim_h = 0, round = 0;
	const i64 wrap = ((i64)1 << s->hash_bits) - 1;
	struct slot *he = &s->tbl[h];

	while (!slot_empty(he)) {
		if (below_next_mask(s, he->t)) { /* due for cleaning: replace */
			s->hash_count--;
			break;
		}
		if (bitness_rank(he->t) < bitness_rank(t)) { /* displace weaker occupant */
			tbl_insert(s, he->t, he->offset);
			break;
		}
		if (he->t == t) {
			if (round == *s->victim_round)
				victim_h = h;
			if (++round == (i64)s->lvl->max_chain_len) {
				h = victim_h;
				he = &s->tbl[h];
				s->hash_count--;
				if (++*s->victim_round == (i64)s->lvl->max_chain_len)
					*s->victim_round = 0;
				break;
			}
		}
		h = (h + 1) & wrap;
		he = &s->tbl[h];
	}
	he->t = t;
	he->offset = offset;
}

static uint64_t tbl_clean_one(struct scan *s)
{
	const i64 size = (i64)1 << s->hash_bits;
	for (;;) {
		uint64_t better = mask_up(s->minimum_tag_mask);
		for (; s->tag_clean_ptr < size; s->tag_clean_ptr++) {
			struct slot *he = &s->tbl[s->tag_clean_ptr];
			if (slot_empty(he))
				continue;
			if ((he->t & better) != better) {
				he->offset = 0;
				he->t = 0;
				s->hash_count--;
				return better;
			}
		}
		s->minimum_tag_mask = better;
		s->tag_clean_ptr = 0;
	}
}

static i64 match_extent(const struct scan *s, i64 p0, i64 op, i64 end, i64 *rev)
{
	const uchar *b = s->buf;
	i64 p = p0, len, floor_p;

	if (op >= p0)
		return 0;
	while (p < end && b[p] == b[op]) {
		p++;
		op++;
	}
	len = p - p0;
	p = p0;
	op -= len;
	floor_p = s->last_match > 0 ? s->last_match : 0;
	while (p > floor_p && op > 0 && b[op - 1] == b[p - 1]) {
		op--;
		p--;
	}
	*rev = p0 - p;
	len += *rev;
	return len < MINIMUM_MATCH ? 0 : len;
}

static i64 tbl_lookup(struct scan *s, uint64_t t, i64 p, i64 end, i64 *offset, i64 *reverse)
{
	const i64 wrap = ((i64)1 << s->hash_bits) - 1;
	i64 h = bucket_of(s, t), best = 0, rev = 0;
	struct slot *he = &s->tbl[h];

	*reverse = 0;
	while (!slot_empty(he)) {
		if (he->t == t) {
			i64 mlen = match_extent(s, p, he->offset, end, &rev);
			if (mlen) {
				if (mlen > best) {
					best = mlen;
					*offset = he->offset - rev;
					*reverse = rev;
				}
				s->st.tag_hits++;
			} else
				s->st.tag_misses++;
		}
		h = (h + 1) & wrap;
		he = &s->tbl[h];
	}
	return best;
}

static uint64_t window_tag(const struct scan *s, i64 p)
{
	uint64_t t = 0;
	int i;
	for (i = 0; i < MINIMUM_MATCH; i++)
		t ^= s->hx[s->buf[p + i]];
	return t;
}

static void put_le(const struct scan *s, uint64_t v, int n)
{
	uchar b[8];
	int i;
	for (i = 0; i < n; i++)
		b[i] = (uchar)(v >> (8 * i));
	s->sink->put0(s->sink->ctx, b, n);
}

static void emit_header(const struct scan *s, uchar head, i64 len)
{
	put_le(s, head, 1);
	put_le(s, (uint64_t)len, 2);
}

static void emit_match(struct scan *s, i64 p, i64 offset, i64 len)
{
	do {
		i64 n = len > 0xFFFF ? 0xFFFF : len;
		emit_header(s, 1, n);
		put_le(s, (uint64_t)(p - offset), s->chunk_bytes);
		s->st.matches++;
		s->st.match_bytes += n;
		len -= n;
		p += n;
		offset += n;
	} while (len);
}

static void emit_literal(struct scan *s, i64 last, i64 p)
{
	do {
		i64 len = p - last;
		if (len > 0xFFFF)
			len = 0xFFFF;
		s->st.literals++;
		s->st.literal_bytes += len;
		emit_header(s, 0, len);
		if (len)
			s->sink->put1(s->sink->ctx, last, len);
		last += len;
	} while (p > last);
}

void lrzo_rzip_chunk_table(const uchar *buf, i64 chunk_size, int rzip_level, int chunk_bytes,
			   const uint64_t hash_index[256], i64 *victim_round,
			   const lrzo_sink *sink, lrzo_rzip_stats *stats, uint32_t *crc_out,
			   uint64_t *table_out)
{
	struct scan s;
	uint64_t t = 0, tag_mask;
	i64 p = 0, end, hashsize;
	struct { i64 p, ofs, len; } cur = {0, 0, 0};

	memset(&s, 0, sizeof(s));
	s.buf = buf;
	s.chunk_size = chunk_size;
	s.chunk_bytes = chunk_bytes;
	s.hx = hash_index;
	s.lvl = &LEVELS[rzip_level];
	s.victim_round = victim_round;
	s.sink = sink;

	hashsize = (i64)s.lvl->mb_used * (1048576 / (i64)sizeof(struct slot));
	for (s.hash_bits = 0; ((i64)1 << s.hash_bits) < hashsize; s.hash_bits++)
		;
	s.hash_limit = ((i64)1 << s.hash_bits) / 3 * 2;
	s.tbl = calloc((size_t)1 << s.hash_bits, sizeof(struct slot));
	if (!s.tbl)
		abort();

	tag_mask = ((uint64_t)1 << s.lvl->initial_freq) - 1;
	s.minimum_tag_mask = tag_mask;
	end = chunk_size - MINIMUM_MATCH;
	if (end > 0)
		t = window_tag(&s, 0);

	while (p < end) {
		i64 reverse, mlen, offset = 0;

		++p;
		t ^= s.hx[buf[p - 1]] ^ s.hx[buf[p + MINIMUM_MATCH - 1]];
		if ((t & s.minimum_tag_mask) != s.minimum_tag_mask)
			continue;

		s.st.lookups++;
		mlen = tbl_lookup(&s, t, p, end, &offset, &reverse);

		if ((t & tag_mask) == tag_mask) {
			s.st.inserts++;
			s.hash_count++;
			tbl_insert(&s, t, p);
			if (s.hash_count > s.hash_limit)
				tag_mask = tbl_clean_one(&s);
		}

		if (mlen > cur.len) {
			cur.p = p - reverse;
			cur.len = mlen;
			cur.ofs = offset;
		}

		if ((cur.len >= GREAT_MATCH || p >= cur.p + MINIMUM_MATCH) && cur.len >= MINIMUM_MATCH) {
			if (s.last_match < cur.p)
				emit_literal(&s, s.last_match, cur.p);
			emit_match(&s, cur.p, cur.ofs, cur.len);
			s.last_match = cur.p + cur.len;
			cur.p = p = s.last_match;
			cur.len = 0;
			if (p < end) /* reference recomputes unconditionally; value is unused when p >= end */
				t = window_tag(&s, p);
		}
	}

	if (s.last_match < chunk_size)
		emit_literal(&s, s.last_match, chunk_size);

	{
		/* src/rzip.c:757-760: terminator, then CRC-32 via put_u32(htole32(cksum)) where
		 * cksum was memcpy'd from gcrypt's big-endian digest => bytes appear MSB first. */
		uint32_t crc = lrzo_crc32(0, buf, (size_t)chunk_size);
		uchar cb[4] = { (uchar)(crc >> 24), (uchar)(crc >> 16), (uchar)(crc >> 8), (uchar)crc };
		emit_literal(&s, 0, 0);
		sink->put0(sink->ctx, cb, 4);
		if (crc_out)
			*crc_out = crc;
	}

	s.st.hash_count = s.hash_count;
	s.st.minimum_tag_mask = s.minimum_tag_mask;
	s.st.tag_mask = tag_mask;
	s.st.tag_clean_ptr = s.tag_clean_ptr;
	if (stats)
		*stats = s.st;
	if (table_out)
		memcpy(table_out, s.tbl, sizeof(struct slot) << s.hash_bits);
	free(s.tbl);
}

void lrzo_rzip_chunk(const uchar *buf, i64 chunk_size, int rzip_level, int chunk_bytes,
		     const uint64_t hash_index[256], i64 *victim_round,
		     const lrzo_sink *sink, lrzo_rzip_stats *stats, uint32_t *crc_out)
{
	lrzo_rzip_chunk_table(buf, chunk_size, rzip_level, chunk_bytes, hash_index, victim_round,
			      sink, stats, crc_out, NULL);
}
