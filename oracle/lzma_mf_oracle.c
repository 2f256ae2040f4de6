/* lzma_mf_oracle.c -- ORACLE (test infrastructure; see lrzo.h).
 *
 * CPU restatement of the match lists the LZMA encoder receives in the reference's
 * default multithreaded mode (numThreads=2, btMode=1, numHashBytes=4):
 *   hash mask                 reference src/lzma/C/LzFind.c:347-373, 432-442
 *   hash thread heads         reference src/lzma/C/LzFindMt.c:368-394 (GetHeads4 / GetHeads4b)
 *   BT thread tree walk       reference src/lzma/C/LzFindMt.c:571-729 + LzFindOpt.c:67-244
 *   LZ thread merge           reference src/lzma/C/LzFindMt.c:1031-1072 (MixMatches3), 1274-1317
 * The long-match shortcut of LzFindOpt.c:163-199 yields the same record and the
 * same son[] pair as a regular walk (first node matches to lenLimit), so it is
 * not restated separately.
 */
#include <stdlib.h>
#include <string.h>
#include "lrzo.h"

uint32_t lrzo_lzma_hash_mask(uint32_t dict_size, uint64_t expected_size)
{
	uint32_t out[2], k;
	uint64_t src[2];
	src[0] = dict_size;
	src[1] = expected_size < dict_size ? expected_size : dict_size;
	for (k = 0; k < 2; k++) {
		uint32_t hs = (uint32_t)src[k];
		if (hs != 0)
			hs--;
		hs |= hs >> 1;
		hs |= hs >> 2;
		hs |= hs >> 4;
		hs |= hs >> 8;
		hs >>= 1;
		if (hs >= (1u << 24))
			hs >>= 1;
		hs |= 0xFFFF;
		out[k] = hs;
	}
	return out[1] > out[0] ? out[0] : out[1];
}

i64 lrzo_lzma_mf_bt4(const uchar *src, size_t n, uint32_t dict_size, unsigned fb, unsigned cut,
		     uint64_t *offsets, uint32_t *pairs, size_t pairs_cap)
{
	uint32_t crc[256], mask, *hash, *son, *h2tab, *h3tab;
	const uint32_t cyc_size = dict_size + 1;
	uint32_t cyc_pos = 1;
	size_t i, nson, total = 0;
	int big;
	uint32_t rec[2 * 300];

	for (i = 0; i < 256; i++) {
		uint32_t r = (uint32_t)i;
		int j;
		for (j = 0; j < 8; j++)
			r = (r >> 1) ^ (0xEDB88320u & (0u - (r & 1)));
		crc[i] = r;
	}
	mask = lrzo_lzma_hash_mask(dict_size, n);
	big = mask >= 0xFFFFFF;
	nson = (n + 2 < cyc_size) ? n + 2 : cyc_size;
	hash = calloc((size_t)mask + 1, 4);
	son = calloc(nson * 2, 4);
	h2tab = calloc(1 << 10, 4);
	h3tab = calloc(1 << 16, 4);
	if (!hash || !son || !h2tab || !h3tab)
		return -1;

	for (i = 0; i < n; i++) {
		const uchar *cur = src + i;
		const uint32_t pos = (uint32_t)(i + 1);
		const size_t avail = n - i;
		unsigned nrec = 0;
		uint32_t *d = rec;

		/* ---- hash + BT threads ---- */
		if (avail >= 4) {
			uint32_t hv, delta, cbs, len_limit = (uint32_t)(avail < fb ? avail : fb);
			if (big)
				hv = (crc[cur[0]] & mask) ^ ((uint32_t)cur[1] | ((uint32_t)cur[2] << 8) | ((uint32_t)cur[3] << 16));
			else
				hv = (crc[cur[0]] & mask) ^ ((crc[cur[3]] << 5) & mask) ^ ((uint32_t)cur[1] | ((uint32_t)cur[2] << 8));
			delta = pos - hash[hv];
			hash[hv] = pos;
			cbs = pos < cyc_size ? pos : cyc_size;
			if (delta >= cbs) {
				son[(size_t)cyc_pos * 2] = 0;
				son[(size_t)cyc_pos * 2 + 1] = 0;
			} else {
				uint32_t *ptr0 = son + (size_t)cyc_pos * 2 + 1, *ptr1 = son + (size_t)cyc_pos * 2;
				uint32_t len0 = 0, len1 = 0, max_len = 3, cv = cut;
				for (;;) {
					uint32_t *pair = son + (((size_t)cyc_pos - delta + (cyc_pos < delta ? cbs : 0)) << 1);
					const uchar *pb = cur - delta;
					uint32_t len = len0 < len1 ? len0 : len1;
					uint32_t pair0 = pair[0];
					if (pb[len] == cur[len]) {
						while (++len != len_limit)
							if (pb[len] != cur[len])
								break;
						if (max_len < len) {
							max_len = len;
							*d++ = len;
							*d++ = delta - 1;
							if (len == len_limit) {
								uint32_t pair1 = pair[1];
								*ptr1 = pair0;
								*ptr0 = pair1;
								break;
							}
						}
					}
					{
						uint32_t cur_match = pos - delta;
						if (pb[len] < cur[len]) {
							delta = pair[1];
							*ptr1 = cur_match;
							ptr1 = pair + 1;
							len1 = len;
						} else {
							delta = pair[0];
							*ptr0 = cur_match;
							ptr0 = pair;
							len0 = len;
						}
						if (delta >= cur_match) { /* corrupt tree: reference returns NULL */
							free(hash); free(son); free(h2tab); free(h3tab);
							return -1;
						}
						delta = pos - delta;
						if (--cv == 0 || delta >= cbs) {
							*ptr0 = *ptr1 = 0;
							break;
						}
					}
				}
			}
			nrec = (unsigned)(d - rec);
		}
		if (++cyc_pos == cyc_size)
			cyc_pos = 0;

		/* ---- LZ thread: MatchFinderMt_GetMatches with MixMatches3 ---- */
		{
			uint32_t out[2 * 4 + 2 * 300], *o = out;
			int do_mix = 0;
			uint32_t min_pos = 0;
			if (nrec == 0) {
				if (avail - 1 >= 3) {
					do_mix = 1;
					min_pos = pos > dict_size ? pos - dict_size : 1;
				}
			} else {
				do_mix = 1;
				min_pos = pos - rec[1];
			}
			if (do_mix) {
				uint32_t temp = crc[cur[0]] ^ cur[1];
				uint32_t h2 = temp & 1023, h3 = (temp ^ ((uint32_t)cur[2] << 8)) & 0xFFFF;
				uint32_t c2 = h2tab[h2], c3 = h3tab[h3];
				int done = 0;
				h2tab[h2] = pos;
				h3tab[h3] = pos;
				if (c2 >= min_pos && cur[(i64)c2 - (i64)pos] == cur[0]) {
					o[1] = pos - c2 - 1;
					if (cur[(i64)c2 - (i64)pos + 2] == cur[2]) {
						o[0] = 3;
						o += 2;
						done = 1;
					} else {
						o[0] = 2;
						o += 2;
					}
				}
				if (!done && c3 >= min_pos && cur[(i64)c3 - (i64)pos] == cur[0]) {
					*o++ = 3;
					*o++ = pos - c3 - 1;
				}
			} else if (avail >= 3) {
				/* Skip() path would still refresh h2/h3 here (LzFindMt.c:1340-1350); it is
				 * unobservable (no later position can read them) so nothing to do. */
			}
			memcpy(o, rec, nrec * 4);
			o += nrec;
			offsets[i] = total;
			if (pairs) {
				size_t cnt = (size_t)(o - out);
				if (total + cnt > pairs_cap) {
					free(hash); free(son); free(h2tab); free(h3tab);
					return -2;
				}
				memcpy(pairs + total, out, cnt * 4);
			}
			total += (size_t)(o - out);
		}
	}
	offsets[n] = total;
	free(hash);
	free(son);
	free(h2tab);
	free(h3tab);
	return (i64)total;
}

/* ---- HC5: the single-threaded hash-chain finder of LZMA levels 1-4 ------------------------------
 * Serial restatement with the reference's own data structures (fixed h2/h3 tables, main 5-byte
 * hash, cyclic chain buffer):
 *   hash mask                    reference src/lzma/C/LzFind.c:347-373 (numHashBytes >= 5: |= 0x3FFFF)
 *   HASH5_CALC                   reference src/lzma/C/LzFind.c:56-63
 *   Hc5_MatchFinder_GetMatches   reference src/lzma/C/LzFind.c:1431-1502
 *   Hc_GetMatchesSpec            reference src/lzma/C/LzFind.c:880-958
 * GetMatches and Skip (LzFind.c:1619-1649) update the tables identically, so the list of a position
 * does not depend on which of the two the parser calls for the positions before it. */
uint32_t lrzo_lzma_hash_mask5(uint32_t dict_size, uint64_t expected_size)
{
	uint32_t m = lrzo_lzma_hash_mask(dict_size, expected_size);
	uint32_t full = lrzo_lzma_hash_mask(dict_size, dict_size);
	m |= (256u << 10) - 1;
	full |= (256u << 10) - 1;
	return m > full ? full : m;
}

i64 lrzo_lzma_mf_hc5(const uchar *src, size_t n, uint32_t dict_size, unsigned fb, unsigned cut,
		     uint64_t *offsets, uint32_t *pairs, size_t pairs_cap)
{
	uint32_t crc[256], mask, *hash5, *son, *h2tab, *h3tab;
	const uint32_t cyc_size = dict_size + 1;
	uint32_t cyc_pos = 0;
	size_t i, total = 0;
	uint32_t out[2 * 70];

	for (i = 0; i < 256; i++) {
		uint32_t r = (uint32_t)i;
		int j;
		for (j = 0; j < 8; j++)
			r = (r >> 1) ^ (0xEDB88320u & (0u - (r & 1)));
		crc[i] = r;
	}
	if (cut > 64)
		return -1;
	mask = lrzo_lzma_hash_mask5(dict_size, n);
	hash5 = calloc((size_t)mask + 1, 4);
	son = calloc((size_t)cyc_size, 4);
	h2tab = calloc(1024, 4);
	h3tab = calloc(65536, 4);
	if (!hash5 || !son || !h2tab || !h3tab) {
		free(hash5); free(son); free(h2tab); free(h3tab);
		return -1;
	}
	for (i = 0; i < n; i++) {
		const uchar *cur = src + i;
		const uint32_t pos = (uint32_t)i + 1; /* the reference starts at pos 1, 0 = empty */
		const size_t avail = n - i;
		const size_t len_limit = avail < fb ? avail : fb;
		uint32_t *d = out;
		offsets[i] = total;
		if (len_limit >= 5) {
			uint32_t temp = crc[cur[0]] ^ cur[1];
			const uint32_t h2 = temp & 1023;
			uint32_t h3, hv, d2, d3, cur_match, mmm, cv = cut;
			size_t max_len = 4;
			int walk = 1;
			temp ^= (uint32_t)cur[2] << 8;
			h3 = temp & 65535;
			temp ^= crc[cur[3]] << 5;
			hv = (temp ^ (crc[cur[4]] << 10)) & mask;
			d2 = pos - h2tab[h2];
			d3 = pos - h3tab[h3];
			cur_match = hash5[hv];
			h2tab[h2] = pos;
			h3tab[h3] = pos;
			hash5[hv] = pos;
			mmm = cyc_size < pos ? cyc_size : pos;
			for (;;) {
				if (d2 < mmm && *(cur - d2) == *cur) {
					d[0] = 2;
					d[1] = d2 - 1;
					d += 2;
					if (*(cur - d2 + 2) == cur[2]) {
					} else if (d3 < mmm && *(cur - d3) == *cur) {
						d[1] = d3 - 1;
						d += 2;
						d2 = d3;
					} else
						break;
				} else if (d3 < mmm && *(cur - d3) == *cur) {
					d[1] = d3 - 1;
					d += 2;
					d2 = d3;
				} else
					break;
				d[-2] = 3;
				if (*(cur - d2 + 3) != cur[3])
					break;
				{
					size_t l = max_len;
					while (l != len_limit && *(cur + l - d2) == cur[l])
						l++;
					max_len = l;
				}
				d[-2] = (uint32_t)max_len;
				if (max_len == len_limit)
					walk = 0;
				break;
			}
			son[cyc_pos] = cur_match;
			if (walk) {
				do {
					uint32_t delta;
					const uchar *pb;
					size_t len;
					if (cur_match == 0)
						break;
					delta = pos - cur_match;
					if (delta >= cyc_size)
						break;
					cur_match = son[cyc_pos - delta + (delta > cyc_pos ? cyc_size : 0)];
					pb = cur - delta;
					if (cur[max_len] != pb[max_len])
						continue;
					for (len = 0; len != len_limit && cur[len] == pb[len]; len++)
						;
					if (len == len_limit) {
						d[0] = (uint32_t)len_limit;
						d[1] = delta - 1;
						d += 2;
						break;
					}
					if (max_len < len) {
						max_len = len;
						d[0] = (uint32_t)len;
						d[1] = delta - 1;
						d += 2;
					}
				} while (--cv);
			}
		}
		/* MatchFinder_MovePos: every position advances the cyclic buffer */
		if (++cyc_pos == cyc_size)
			cyc_pos = 0;
		if (pairs) {
			size_t cnt = (size_t)(d - out);
			if (total + cnt > pairs_cap) {
				free(hash5); free(son); free(h2tab); free(h3tab);
				return -2;
			}
			memcpy(pairs + total, out, cnt * 4);
		}
		total += (size_t)(d - out);
	}
	offsets[n] = total;
	free(hash5);
	free(son);
	free(h2tab);
	free(h3tab);
	return (i64)total;
}
