// api_hash.cpp -- host-only entry points of include/lrzgpu_hash.h: whole-file hashes, trailer rewrite, read_magic.
#include <cstdlib>
#include <cstring>
#include <new>

#include "../../include/lrzgpu.h"
#include "../../include/lrzgpu_hash.h"
#include "filters.h"
#include "hashes.h"

using namespace lrzgpu;

// ---- scan access hooks: control->full_tag / next_tag / match_len of the non-sliding mode (src/rzip.c:385-393,
// 405-416, 431-461; installed at 1036-1039) over a plain buffer -- the host-side meaning of what k_tag_scan and the
// resolver's match verification compute for whole chunks
extern "C" void lrzgpu_hash_index(uint64_t out[256]);
namespace {
const uint64_t *tag_table()
{
	static uint64_t t[256];
	static const bool once = (lrzgpu_hash_index(t), true);
	(void)once;
	return t;
}
} // namespace
extern "C" uint64_t lrzgpu_full_tag(const uint8_t *buf, int64_t p)
{
	const uint64_t *hx = tag_table();
	uint64_t t = 0;
	for (int i = 0; i < 31; i++) // MINIMUM_MATCH
		t ^= hx[buf[p + i]];
	return t;
}
extern "C" uint64_t lrzgpu_next_tag(const uint8_t *buf, int64_t p, uint64_t t)
{
	const uint64_t *hx = tag_table();
	return t ^ hx[buf[p - 1]] ^ hx[buf[p + 31 - 1]];
}
extern "C" int64_t lrzgpu_match_len(const uint8_t *buf, int64_t p0, int64_t op, int64_t end, int64_t last_match, int64_t *rev)
{
	if (rev)
		*rev = 0;
	if (op >= p0)
		return 0;
	int64_t fwd = 0;
	while (p0 + fwd < end && buf[p0 + fwd] == buf[op + fwd])
		fwd++;
	const int64_t floor_p = last_match > 0 ? last_match : 0;
	int64_t back = 0;
	while (p0 - back > floor_p && op - back > 0 && buf[op - back - 1] == buf[p0 - back - 1])
		back++;
	if (rev)
		*rev = back;
	return fwd + back < 31 ? 0 : fwd + back;
}

// ---- filters (src/stream.c:1587-1628 / 1926-1990) ---------------------------------------------------------------
extern "C" int lrzgpu_filter_supported(int filter_flag, int delta) { return filter_supported(filter_flag, delta) ? 1 : 0; }
extern "C" int lrzgpu_filter_block(int filter_flag, int delta, uint8_t *data, int64_t n, int encode)
{
	if (n < 0 || (n && !data))
		return LRZGPU_E_PARAM;
	return filter_block(filter_flag, delta, data, (size_t)n, encode != 0) == 0 ? 0 : LRZGPU_E_PARAM;
}
// magic[16] of an image (0.13+ coding: 128 + code of the delta distance, else the flag), trailer untouched
extern "C" int lrzgpu_set_file_filter(uint8_t *lrz, int64_t n, int filter_flag, int delta)
{
	if (!lrz || n < 21 || memcmp(lrz, "LRZI", 4) != 0 || lrz[4] != 0 || lrz[5] < 13)
		return LRZGPU_E_PARAM;
	if (filter_flag == FILTER_DELTA) {
		if (delta < 1 || delta > 256 || (delta > 16 && delta % 16))
			return LRZGPU_E_PARAM;
		lrz[16] = (uint8_t)(128 + (delta <= 16 ? delta : delta / 16 + 15)); // write_magic, src/lrzip.c:148-156
	} else if (filter_flag >= 0 && filter_flag <= 8)
		lrz[16] = (uint8_t)filter_flag;
	else
		return LRZGPU_E_PARAM;
	return 0;
}

extern "C" int lrzgpu_hash_length(int hash_code) { return hash_length(hash_code); }
extern "C" const char *lrzgpu_hash_label(int hash_code) { return hash_label(hash_code); }

extern "C" void *lrzgpu_hash_open(int hash_code)
{
	try {
		return make_hasher(hash_code).release();
	} catch (...) {
		return nullptr;
	}
}
extern "C" int lrzgpu_hash_update(void *h, const uint8_t *data, int64_t n)
{
	if (!h || n < 0 || (n && !data))
		return LRZGPU_E_PARAM;
	static_cast<Hasher *>(h)->update(data, (size_t)n);
	return 0;
}
extern "C" int lrzgpu_hash_final(void *h, uint8_t *out)
{
	if (!h)
		return LRZGPU_E_PARAM;
	Hasher *p = static_cast<Hasher *>(h);
	if (out)
		p->finish(out);
	delete p;
	return out ? 0 : LRZGPU_E_PARAM;
}
extern "C" int lrzgpu_hash_buffer(int hash_code, const uint8_t *data, int64_t n, uint8_t *out)
{
	if (n < 0 || (n && !data) || !out)
		return LRZGPU_E_PARAM;
	void *h = lrzgpu_hash_open(hash_code);
	if (!h)
		return LRZGPU_E_PARAM;
	(void)lrzgpu_hash_update(h, data, n);
	return lrzgpu_hash_final(h, out);
}

// magic[14] and the bytes after the last chunk (src/lrzip.c:146, src/rzip.c:1195-1219)
extern "C" int lrzgpu_set_file_hash(const uint8_t *lrz, int64_t n, int hash_code, const uint8_t *digest, uint8_t **out, int64_t *out_len)
{
	if (!lrz || !out || !out_len || n < 21)
		return LRZGPU_E_PARAM;
	const int new_len = hash_code == 0 ? 0 : hash_length(hash_code);
	if (new_len < 0 || (new_len && !digest))
		return LRZGPU_E_PARAM;
	if (memcmp(lrz, "LRZI", 4) != 0 || lrz[4] != 0 || lrz[5] < 11 || lrz[5] > 14 || lrz[15])
		return LRZGPU_E_FORMAT;
	const int old_len = lrz[14] == 0 ? 0 : hash_length(lrz[14]);
	if (old_len < 0 || n < 21 + (int64_t)lrz[20] + old_len)
		return LRZGPU_E_FORMAT;
	const int64_t body = n - old_len;
	uint8_t *o = (uint8_t *)malloc((size_t)(body + new_len));
	if (!o)
		return LRZGPU_E_NOMEM;
	memcpy(o, lrz, (size_t)body);
	o[14] = (uint8_t)hash_code;
	if (new_len)
		memcpy(o + body, digest, (size_t)new_len);
	*out = o;
	*out_len = body + new_len;
	return 0;
}

namespace {
uint64_t le64(const uint8_t *p)
{
	uint64_t v = 0;
	for (int i = 7; i >= 0; i--)
		v = v << 8 | p[i];
	return v;
}
uint32_t lzma2_dict_from_prop(unsigned p) { return p == 40 ? 0xFFFFFFFFu : ((uint32_t)2 | (p & 1)) << (p / 2 + 11); } // LZMA2_DIC_SIZE_FROM_PROP
void take_hash(lrzgpu_magic *m, unsigned v) // get_hash_from_magic, src/lrzip.c:248-262
{
	if (v > 0 && v <= LRZGPU_HASH_MAX) {
		m->hash_code = (int)v;
		m->hash_len = hash_length((int)v);
	}
}
void take_encryption(lrzgpu_magic *m, unsigned v, const uint8_t *salt) // get_encryption, src/lrzip.c:266-290
{
	if (v > 0 && v <= 2) {
		m->enc_code = (int)v;
		memcpy(m->salt, salt, 8);
		m->costfactor = salt[0];
		m->st_size = 0;
	}
}
void take_filter(lrzgpu_magic *m, unsigned v) // get_filter, src/lrzip.c:304-338: three encodings of the delta distance
{
	if (!v)
		return;
	if (m->minor < 12) {
		m->filter_flag = (int)(v & 7);
		if (m->filter_flag == 7) { // OLD_FILTER_FLAG_DELTA: offset stored as value - 1 in the high five bits
			const int i = (int)((v & 0xF8) >> 3);
			m->filter_flag = 128;
			m->delta = i <= 16 ? i + 1 : (i - 16 + 1) * 16;
		}
	} else if (m->minor == 12) {
		if (v & 0xF8) {
			const int i = (int)(v >> 3);
			m->filter_flag = 128;
			m->delta = i <= 16 ? i : (i - 15) * 16;
		} else
			m->filter_flag = (int)v;
	} else {
		if (v > 128) {
			const int i = (int)v - 128;
			m->filter_flag = 128;
			m->delta = i <= 16 ? i : (i - 15) * 16;
		} else
			m->filter_flag = (int)v;
	}
}
void lzma_from_dict_prop(lrzgpu_magic *m, unsigned prop)
{
	m->ctype = 1;
	m->dict_size = lzma2_dict_from_prop(prop);
	m->lzma_properties[0] = 0x5D; // LZMA_LC_LP_PB
	for (int i = 0; i < 4; i++)
		m->lzma_properties[1 + i] = (uint8_t)(m->dict_size >> (8 * i));
}
} // namespace

extern "C" int lrzgpu_read_magic(const uint8_t *g, int64_t n, lrzgpu_magic *m)
{
	if (!g || !m || n < 6)
		return LRZGPU_E_PARAM;
	memset(m, 0, sizeof(*m));
	if (memcmp(g, "LRZI", 4) != 0)
		return LRZGPU_E_FORMAT;
	m->major = g[4];
	m->minor = g[5];
	if (m->major != 0)
		return LRZGPU_E_FORMAT;
	switch (m->minor) { // read_magic, src/lrzip.c:539-585
	case 6:
	case 7: m->magic_len = 24; break;
	case 8: m->magic_len = 18; break;
	case 9:
	case 10: m->magic_len = 20; break;
	case 11:
	case 12:
	case 13:
	case 14: m->magic_len = 21; break;
	default: return LRZGPU_E_FORMAT;
	}
	if (n < m->magic_len)
		return LRZGPU_E_FORMAT;
	int comment_at = -1;
	if (m->minor == 6) { // get_magic_v6
		if (!g[22])
			m->st_size = (int64_t)le64(g + 6);
		if (g[16]) {
			m->ctype = 1;
			memcpy(m->lzma_properties, g + 16, 5);
			m->dict_size = (uint32_t)g[17] | (uint32_t)g[18] << 8 | (uint32_t)g[19] << 16 | (uint32_t)g[20] << 24;
		}
		take_hash(m, g[21]);
		take_encryption(m, g[22], g + 6);
	} else if (m->minor == 7) { // get_magic_v7
		if (!g[23])
			m->st_size = (int64_t)le64(g + 6);
		take_encryption(m, g[23], g + 6);
		take_filter(m, g[16]);
		if (g[17]) {
			m->ctype = 1;
			memcpy(m->lzma_properties, g + 17, 5);
			m->dict_size = (uint32_t)g[18] | (uint32_t)g[19] << 8 | (uint32_t)g[20] << 16 | (uint32_t)g[21] << 24;
		}
		take_hash(m, g[22]);
	} else if (m->minor <= 10) { // get_magic_v8 (+ get_magic_v9)
		if (!g[15])
			m->st_size = (int64_t)le64(g + 6);
		take_encryption(m, g[15], g + 6);
		take_filter(m, g[16]);
		if (g[17] > 0 && g[17] <= 40)
			lzma_from_dict_prop(m, g[17]);
		else if (g[17] & 0x80) {
			if ((g[17] & 0xF0) == 0xF0) {
				m->ctype = 3;
				m->bzip3_bs = g[17] & 0x0F;
			} else {
				m->ctype = 2;
				m->zpaq_bs = g[17] & 0x0F;
				m->zpaq_level = (g[17] & 0x70) >> 4;
			}
		}
		take_hash(m, g[14]);
		if (m->minor >= 9) {
			m->level = g[18] & 0x0F;
			m->rzip_level = g[18] >> 4;
			comment_at = 19;
		}
	} else { // get_magic_v11
		if (!g[15])
			m->st_size = (int64_t)le64(g + 6);
		take_encryption(m, g[15], g + 6);
		take_filter(m, g[16]);
		if (g[17] == 1)
			lzma_from_dict_prop(m, g[18]);
		else if (g[17] == 2) {
			m->ctype = 2;
			m->zpaq_bs = g[18] & 0x0F;
			m->zpaq_level = g[18] >> 4;
		} else if (g[17] == 3) {
			m->ctype = 3;
			m->bzip3_bs = g[18] & 0x0F;
		} else if ((g[17] & 0x0F) == 4) {
			m->ctype = 4;
			m->zstd_strategy = g[17] >> 4;
			m->zstd_level = g[18];
		} else if (g[17] != 0)
			return LRZGPU_E_FORMAT; // "Invalid compression type"
		take_hash(m, g[14]);
		m->level = g[19] & 0x0F;
		m->rzip_level = g[19] >> 4;
		comment_at = 20;
	}
	if (comment_at >= 0 && g[comment_at]) { // get_comment
		m->comment_length = g[comment_at];
		if (n < m->magic_len + m->comment_length)
			return LRZGPU_E_FORMAT;
		memcpy(m->comment, g + m->magic_len, (size_t)m->comment_length);
	}
	return 0;
}
