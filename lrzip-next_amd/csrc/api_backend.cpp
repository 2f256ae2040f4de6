// api_backend.cpp -- C ABI entry points of the per-block backends (lz4 gate, LZMA).
// Boundary declared in include/lrzgpu.h; each function cites the reference interface it replaces.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <memory>
#include <mutex>
#include <vector>

#include "../../include/lrzgpu.h"
#include "common.h"
#include "lz4_gate.h"
#include "rzip_census.h"
#include "lzma_enc.h"
#include "lzma_mf.h"
#include "filters.h"
#include "filters_gpu.h"
#include "pools.h"
#include "profile.h"

using namespace lrzgpu;

extern "C" int lrzgpu_device_count(void)
{
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess)
		return 0;
	return n;
}

extern "C" const char *lrzgpu_version(void) { return "lrzgpu 0.1 (gfx950; lrzip-next 0.14.0 container)"; }

namespace lrzgpu {
int select_device(int device)
{
	int n = lrzgpu_device_count();
	if (n <= 0 || device < 0 || device >= n)
		return LRZGPU_E_NODEVICE;
	if (hipSetDevice(device) != hipSuccess)
		return LRZGPU_E_HIP;
	return 0;
}
} // namespace lrzgpu

// ---- lz4 gate: src/stream.c:2325-2380 -------------------------------------------------------

// The per-block entry points are called block after block from the same compthreads (src/stream.c:1633-1648):
// each calling thread keeps its device buffers between calls (taken from / returned to the process-wide pools
// when the thread ends); nothing is hipMalloc()ed or hipFree()d per call.
struct ThreadBuffers {
	DevBuf small, block;
	int device = -1;
	hipStream_t s = nullptr; // own non-blocking stream: a null-stream call would wait for every blocking stream of the process
	~ThreadBuffers() { StreamPool::get().give(s); } // parked, never destroyed (pools.h)
	bool ensure(int dev, size_t block_bytes)
	{
		if (device != dev) {
			small.release();
			block.release();
			StreamPool::get().give(s);
			s = nullptr;
			device = dev;
		}
		if (!s && !(s = pooled_stream(dev)))
			return false;
		if (!small.p && !small.alloc(256, dev))
			return false;
		if (block_bytes && block.cap < block_bytes && !block.alloc(block_bytes, dev))
			return false;
		return true;
	}
};
static ThreadBuffers &thread_buffers()
{
	thread_local ThreadBuffers b;
	return b;
}

static int lz4_size_dev(const uint8_t *d_src, int src_size, int dst_capacity, int stop_below = 0)
{
	int dev = 0;
	(void)hipGetDevice(&dev);
	ThreadBuffers &tb = thread_buffers();
	if (!tb.ensure(dev, 0))
		return LRZGPU_E_NOMEM;
	Lz4Job job{d_src, src_size, dst_capacity, stop_below}, *d_job = (Lz4Job *)tb.small.p;
	int *d_res = (int *)(tb.small.p + 64), res = -1;
	if (hipMemcpyAsync(d_job, &job, sizeof(job), hipMemcpyHostToDevice, tb.s) != hipSuccess || lz4_sizes_device(d_job, 1, d_res, tb.s) != 0 ||
	    hipMemcpyAsync(&res, d_res, sizeof(int), hipMemcpyDeviceToHost, tb.s) != hipSuccess || stream_wait(tb.s) != hipSuccess)
		return LRZGPU_E_HIP;
	return res;
}

extern "C" int lrzgpu_lz4_compresses_dev(const void *d_buf, int64_t s_len, int threshold, int device)
{
	int rc = select_device(device);
	if (rc)
		return rc;
	if (s_len < 0)
		return LRZGPU_E_PARAM;
	int err = 0;
	int v = lz4_compresses_decision(s_len, threshold, [&](int in_len, int d_len) {
		int r = lz4_size_dev((const uint8_t *)d_buf, in_len, d_len);
		if (r < 0) {
			err = r;
			return 0;
		}
		return r;
	});
	return err ? err : v;
}

// the duplicate census of the scan (rzip_census.hip) on its own: 1 = no 31-byte window of s_buf[0..s_len) occurs twice
// (exact), 0 = some may; stats (may be NULL) = sample anchors, equal neighbours in the sample, all anchors, equal values among them, those of them that were chance
extern "C" int lrzgpu_census(const uint8_t *s_buf, int64_t s_len, int device, int64_t stats[5])
{
	int rc = select_device(device);
	if (rc)
		return rc;
	if (s_len < 0)
		return LRZGPU_E_PARAM;
	ThreadBuffers &tb = thread_buffers();
	if (!tb.ensure(device, (size_t)s_len + 16))
		return LRZGPU_E_NOMEM;
	if (s_len && (hipMemcpyAsync(tb.block.p, s_buf, (size_t)s_len, hipMemcpyHostToDevice, tb.s) != hipSuccess || stream_wait(tb.s) != hipSuccess))
		return LRZGPU_E_HIP;
	CensusStats cs;
	const int v = duplicate_census(tb.block.p, s_len, device, tb.s, &cs);
	if (stats) {
		stats[0] = cs.sample_anchors;
		stats[1] = cs.sample_equal;
		stats[2] = cs.anchors;
		stats[3] = cs.equal;
		stats[4] = cs.cleared;
	}
	return v < 0 ? LRZGPU_E_HIP : v;
}

extern "C" int lrzgpu_lz4_compresses(const uint8_t *s_buf, int64_t s_len, int threshold, int device)
{
	int rc = select_device(device);
	if (rc)
		return rc;
	if (s_len < 0)
		return LRZGPU_E_PARAM;
	ThreadBuffers &tb = thread_buffers();
	if (!tb.ensure(device, (size_t)s_len + 16))
		return LRZGPU_E_NOMEM;
	if (s_len && (hipMemcpyAsync(tb.block.p, s_buf, (size_t)s_len, hipMemcpyHostToDevice, tb.s) != hipSuccess || stream_wait(tb.s) != hipSuccess))
		return LRZGPU_E_HIP;
	return lrzgpu_lz4_compresses_dev(tb.block.p, s_len, threshold, device);
}

static int lz4_size_host(const uint8_t *src, int src_size, int dst_capacity, int stop_below, int device);

extern "C" int lrzgpu_lz4_compress_default_size(const uint8_t *src, int src_size, int dst_capacity, int device)
{
	return lz4_size_host(src, src_size, dst_capacity, 0, device);
}

extern "C" int lrzgpu_lz4_size_stop_below(const uint8_t *src, int src_size, int dst_capacity, int stop_below, int device)
{
	return lz4_size_host(src, src_size, dst_capacity, stop_below > 0 ? stop_below : 0, device);
}

static int lz4_size_host(const uint8_t *src, int src_size, int dst_capacity, int stop_below, int device)
{
	int rc = select_device(device);
	if (rc)
		return rc;
	if (src_size < 0)
		return 0;
	ThreadBuffers &tb = thread_buffers();
	if (!tb.ensure(device, (size_t)src_size + 16))
		return LRZGPU_E_NOMEM;
	if (src_size && (hipMemcpyAsync(tb.block.p, src, (size_t)src_size, hipMemcpyHostToDevice, tb.s) != hipSuccess || stream_wait(tb.s) != hipSuccess))
		return LRZGPU_E_HIP;
	return lz4_size_dev(tb.block.p, src_size, dst_capacity, stop_below);
}

// ---- LZMA: src/lzma/include/LzmaLib.h:95-112 --------------------------------------------------

static int64_t match_lists_impl(const uint8_t *src, size_t n, uint32_t dictSize, unsigned fb, unsigned cutValue,
				uint8_t *counts, uint32_t *pairs, size_t pairs_cap, int device, bool hc5, size_t block_n = 0)
{
	int rc = select_device(device);
	if (rc)
		return rc;
	double per_pos = n ? (double)pairs_cap / (double)n : 16.0, got_per_pos = 0;
	if (per_pos < 4)
		per_pos = 4;
	MfWorkspace *w = WorkspacePool::get().take_mf(n ? n : 1, per_pos, device, &got_per_pos);
	if (!w)
		return LRZGPU_E_NOMEM;
	ThreadBuffers &tb = thread_buffers();
	int64_t ret = LRZGPU_E_HIP;
	unsigned long long total = 0;
	if (!tb.ensure(device, n + 16))
		ret = LRZGPU_E_NOMEM;
	else if (n == 0 || hipMemcpy(tb.block.p, src, n, hipMemcpyHostToDevice) == hipSuccess) {
		const uint8_t *d_src = tb.block.p;
		int r = mf_run_device(w, d_src, n, dictSize, fb, cutValue, 0, &total, 0, hc5, block_n);
		if (r == 0) {
			if (total > pairs_cap)
				ret = LRZGPU_E_NOMEM;
			else if ((n == 0 || hipMemcpy(counts, w->counts, n, hipMemcpyDeviceToHost) == hipSuccess) &&
				 (total == 0 || hipMemcpy(pairs, w->pool_out, total * 4, hipMemcpyDeviceToHost) == hipSuccess))
				ret = (int64_t)total;
		} else
			ret = r == -4 ? LRZGPU_E_NOMEM : LRZGPU_E_INTERNAL;
	}
	WorkspacePool::get().give_mf(w, got_per_pos, device);
	return ret;
}

extern "C" int64_t lrzgpu_lzma_match_lists(const uint8_t *src, size_t n, uint32_t dictSize, unsigned fb, unsigned cutValue,
					   uint8_t *counts, uint32_t *pairs, size_t pairs_cap, int device)
{
	return match_lists_impl(src, n, dictSize, fb, cutValue, counts, pairs, pairs_cap, device, false);
}

extern "C" int64_t lrzgpu_lzma_match_lists_prefix(const uint8_t *src, size_t n, size_t block_n, uint32_t dictSize, unsigned fb,
						  unsigned cutValue, uint8_t *counts, uint32_t *pairs, size_t pairs_cap, int device)
{
	if (block_n < n)
		return LRZGPU_E_PARAM;
	return match_lists_impl(src, n, dictSize, fb, cutValue, counts, pairs, pairs_cap, device, false, block_n);
}

extern "C" int64_t lrzgpu_lzma_match_lists_hc5(const uint8_t *src, size_t n, uint32_t dictSize, unsigned fb, unsigned cutValue,
					       uint8_t *counts, uint32_t *pairs, size_t pairs_cap, int device)
{
	return match_lists_impl(src, n, dictSize, fb, cutValue, counts, pairs, pairs_cap, device, true);
}

// ---- streaming finder in the BT thread's block format (LzFindMt.c:39-42, 571-729) ------------------------------
struct lrzgpu_mf {
	std::vector<uint8_t> counts;
	std::vector<uint32_t> pairs;
	size_t n = 0, pos = 0, off = 0;
	unsigned fb = 0;
};

extern "C" int lrzgpu_lzma_mf_open(lrzgpu_mf **mf, const uint8_t *src, size_t n, uint32_t dictSize, unsigned fb, unsigned cutValue, int device)
{
	if (!mf || (!src && n) || fb < 5 || fb > 273)
		return LRZGPU_E_PARAM;
	try {
		std::unique_ptr<lrzgpu_mf> m(new lrzgpu_mf());
		m->n = n;
		m->fb = fb;
		m->counts.resize(n ? n : 1);
		size_t cap = n * 16 + 4096;
		for (int attempt = 0;; attempt++) {
			m->pairs.resize(cap);
			const int64_t total = match_lists_impl(src, n, dictSize, fb, cutValue, m->counts.data(), m->pairs.data(), cap, device, false);
			if (total == LRZGPU_E_NOMEM && attempt < 3) {
				cap *= 4;
				continue;
			}
			if (total < 0)
				return (int)total;
			m->pairs.resize((size_t)total);
			break;
		}
		*mf = m.release();
		return 0;
	} catch (...) {
		return LRZGPU_E_NOMEM;
	}
}

extern "C" int lrzgpu_lzma_mf_next_block(lrzgpu_mf *m, uint32_t *d, size_t cap_u32)
{
	constexpr uint32_t kBlock = 1u << 16; // kMtBtBlockSize
	if (!m || !d || cap_u32 < kBlock)
		return LRZGPU_E_PARAM;
	if (m->pos >= m->n) {
		d[0] = 2;
		d[1] = 0;
		return 0;
	}
	const uint32_t limit = kBlock - m->fb * 2; // a record may run past it by one list: that is what the slack is for
	uint32_t cur = 2;
	const size_t first = m->pos;
	const size_t left = m->n - m->pos;
	d[1] = left > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)left; // bytes available at the block's first position
	while (cur < limit && m->pos < m->n) {
		const unsigned c = m->counts[m->pos];
		const uint32_t *p = m->pairs.data() + m->off;
		// the LZ thread's h2/h3 candidates (lengths 2 and 3, MixMatches3) come first in the merged list; the BT
		// thread's own records start at length 4 (maxLen starts at numHashBytes - 1)
		unsigned k = 0;
		while (k < c && p[k] < 4)
			k += 2;
		d[cur++] = c - k;
		for (; k < c; k++)
			d[cur++] = p[k];
		m->off += c;
		m->pos++;
	}
	d[0] = cur;
	return (int)(m->pos - first);
}

extern "C" void lrzgpu_lzma_mf_close(lrzgpu_mf *m) { delete m; }

// The parser on lists that arrive in stages (lzma_enc.h StagedLists; DESIGN.md section 5 "early start").  This
// host-only entry is the harness of that path: it runs the encoder on a PRIVATE copy of the block whose bytes from
// the current limit on are 0xA5 and on lists cut off at that limit; every time the parser asks for more, stage_step
// further positions (0 = all that is left) are revealed in place -- an encoder that looked beyond what a stage
// covers would not produce the whole-block stream.
namespace {
struct StagedHarness {
	std::vector<uint8_t> block;
	std::vector<uint8_t> ec;
	std::vector<uint32_t> ep;
	const unsigned char *src;
	const uint8_t *counts;
	const uint32_t *pairs;
	size_t n, limit, step, words_done;
	int fmt;
	MatchLists ml;
	int calls = 0;
	static const MatchLists *rest(void *ctx, size_t *valid)
	{
		StagedHarness *h = (StagedHarness *)ctx;
		h->calls++;
		const size_t next = (h->step == 0 || h->limit + h->step > h->n) ? h->n : h->limit + h->step;
		memcpy(h->block.data() + h->limit, h->src + h->limit, next - h->limit);
		uint64_t before = 0, entries = 0;
		for (size_t i = 0; i < h->limit; i++)
			before += h->counts[i];
		for (size_t i = h->limit; i < next; i++)
			entries += h->counts[i];
		const size_t w0 = (size_t)(h->fmt == 2 ? before / 2 : before), w = (size_t)(h->fmt == 2 ? entries / 2 : entries);
		memcpy(h->ec.data() + h->limit, h->counts + h->limit, next - h->limit);
		memcpy(h->ep.data() + w0, h->pairs + w0, w * 4);
		h->limit = next;
		*valid = next;
		return &h->ml;
	}
};
} // namespace

extern "C" int lrzgpu_lzma_encode_with_lists_staged(unsigned char *dest, size_t *destLen, const unsigned char *src, size_t srcLen,
						    const uint8_t *counts, const uint32_t *pairs, size_t early_positions, int list_format,
						    int level, unsigned dictSize, int lc, int lp, int pb, int fb,
						    const uint8_t *early_counts, const uint32_t *early_pairs, size_t stage_step)
{
	if (list_format < 0 || list_format > 2 || !dest || !destLen || (!src && srcLen) || !counts || !pairs)
		return LZ_ERROR_PARAM;
	if (list_format == 2 && (dictSize > (1u << 25) || fb > 65))
		return LZ_ERROR_PARAM;
	LzmaParams p;
	p.level = level;
	p.dict_size = dictSize;
	p.lc = lc;
	p.lp = lp;
	p.pb = pb;
	p.fb = fb;
	p.fast = level >= 0 && level < 5;
	try {
		StagedHarness h;
		h.src = src;
		h.counts = counts;
		h.pairs = pairs;
		h.n = srcLen;
		h.step = stage_step;
		h.fmt = list_format;
		h.limit = early_positions < srcLen ? early_positions : srcLen;
		h.block.assign(src, src + h.limit);
		h.block.resize(srcLen + 16, 0xA5);
		// the early stage's own arrays: the lists of the first `early` positions -- the caller's (a finder run on a
		// prefix of the block) or the whole block's --, nothing behind them
		const uint8_t *c0 = early_counts ? early_counts : counts;
		const uint32_t *p0 = early_pairs ? early_pairs : pairs;
		uint64_t entries = 0, all = 0;
		for (size_t i = 0; i < h.limit; i++)
			entries += c0[i];
		for (size_t i = 0; i < srcLen; i++)
			all += counts[i];
		const size_t words = (size_t)(list_format == 2 ? entries / 2 : entries);
		h.ec.assign(c0, c0 + h.limit);
		h.ep.assign(p0, p0 + words);
		h.ec.resize(srcLen + 16, 0xFE); // (what a run-away reader would take for long lists)
		h.ep.resize((size_t)(list_format == 2 ? all / 2 : all) + 4096, 0x7FFFFFFFu);
		h.ml.counts = h.ec.data();
		h.ml.pairs = h.ep.data();
		h.ml.tail_flags = list_format != 0;
		h.ml.packed = list_format == 2;
		StagedLists sl;
		sl.early = h.ml;
		sl.early_positions = h.limit;
		sl.rest = &StagedHarness::rest;
		sl.ctx = &h;
		size_t out_len = 0;
		const size_t first = h.limit;
		const int r = lzma_encode_block_staged(p, h.block.data(), srcLen, sl, dest, *destLen, &out_len);
		*destLen = out_len;
		if (r == LZ_OK && first < srcLen && (h.limit != srcLen || (stage_step == 0 && h.calls != 1)))
			return LZ_ERROR_PARAM; // the rest must have been asked for (once, when it comes in one piece)
		return r;
	} catch (...) {
		return LZ_ERROR_MEM;
	}
}

extern "C" int lrzgpu_lzma_encode_with_lists_fmt(unsigned char *dest, size_t *destLen, const unsigned char *src, size_t srcLen,
						 const uint8_t *counts, const uint32_t *pairs, int list_format, int level, unsigned dictSize,
						 int lc, int lp, int pb, int fb)
{
	if (list_format < 0 || list_format > 2)
		return LZ_ERROR_PARAM;
	LzmaParams p;
	p.level = level;
	p.dict_size = dictSize;
	p.lc = lc;
	p.lp = lp;
	p.pb = pb;
	p.fb = fb;
	p.fast = level >= 0 && level < 5; // algo 0
	if (list_format == 2 && (dictSize > (1u << 25) || fb > 65))
		return LZ_ERROR_PARAM;
	MatchLists ml;
	ml.counts = counts;
	ml.pairs = pairs;
	ml.tail_flags = list_format != 0;
	ml.packed = list_format == 2;
	size_t out_len = 0;
	int r = lzma_encode_block(p, src, srcLen, ml, dest, *destLen, &out_len);
	*destLen = out_len;
	return r;
}

extern "C" int lrzgpu_lzma_encode_with_lists(unsigned char *dest, size_t *destLen, const unsigned char *src, size_t srcLen,
					     const uint8_t *counts, const uint32_t *pairs, int level, unsigned dictSize,
					     int lc, int lp, int pb, int fb)
{
	LzmaParams p;
	p.level = level;
	p.dict_size = dictSize;
	p.lc = lc;
	p.lp = lp;
	p.pb = pb;
	p.fb = fb;
	p.fast = level >= 0 && level < 5; // algo 0
	MatchLists ml;
	ml.counts = counts;
	ml.pairs = pairs;
	size_t out_len = 0;
	int r = lzma_encode_block(p, src, srcLen, ml, dest, *destLen, &out_len);
	*destLen = out_len;
	return r;
}

namespace lrzgpu {
// LzmaEncProps_Normalize() for the arguments lrzip-next passes (LzmaEnc.c:68-108)
int lzma_normalize(LzmaParams &p, int level, unsigned dictSize, int lc, int lp, int pb, int fb)
{
	if (level < 0)
		level = 5;
	p.level = level;
	if (dictSize == 0)
		dictSize = level <= 3 ? (1u << (level * 2 + 16)) : level <= 6 ? (1u << (level + 19)) : level <= 7 ? (1u << 25) : (1u << 26);
	p.dict_size = dictSize;
	p.lc = lc < 0 ? 3 : lc;
	p.lp = lp < 0 ? 0 : lp;
	p.pb = pb < 0 ? 2 : pb;
	p.fb = fb < 0 ? (level < 7 ? 32 : 64) : fb;
	if (p.lc > 8 || p.lp > 4 || p.pb > 4)
		return LZ_ERROR_PARAM;
	p.fast = level < 5; // algo 0: HC5 finder + GetOptimumFast (LzmaEnc.c:95-99)
	return LZ_OK;
}
} // namespace lrzgpu

extern "C" int lrzgpu_LzmaCompress(unsigned char *dest, size_t *destLen, const unsigned char *src, size_t srcLen,
				   unsigned char *outProps, size_t *outPropsSize, int level, unsigned dictSize,
				   int lc, int lp, int pb, int fb, int numThreads)
{
	(void)numThreads;
	LzmaParams p;
	int r = lzma_normalize(p, level, dictSize, lc, lp, pb, fb);
	if (r != LZ_OK)
		return r;
	if (*outPropsSize < 5)
		return LZ_ERROR_PARAM;
	*outPropsSize = 5;
	lzma_write_props(p, outProps);
	int dev = 0;
	(void)hipGetDevice(&dev);
	if (select_device(dev))
		return LZ_ERROR_MEM; // no SRes for "no device": the library has no CPU finder
	std::vector<uint8_t> counts(srcLen ? srcLen : 1);
	size_t cap = srcLen * 16 + 4096;
	for (int attempt = 0; attempt < 3; attempt++) {
		std::vector<uint32_t> pairs;
		try {
			pairs.resize(cap);
		} catch (...) {
			return LZ_ERROR_MEM;
		}
		int64_t total = match_lists_impl(src, srcLen, p.dict_size, (unsigned)p.fb, p.cut(), counts.data(), pairs.data(), cap, dev, p.fast);
		if (total == LRZGPU_E_NOMEM) {
			cap *= 4;
			continue;
		}
		if (total < 0)
			return LZ_ERROR_MEM;
		MatchLists ml;
		ml.counts = counts.data();
		ml.pairs = pairs.data();
		size_t out_len = 0;
		r = lzma_encode_block(p, src, srcLen, ml, dest, *destLen, &out_len);
		*destLen = out_len;
		return r;
	}
	return LZ_ERROR_MEM;
}

extern "C" void lrzgpu_profile_reset(void)
{
	ProfileStore &ps = ProfileStore::get();
	std::lock_guard<std::mutex> lk(ps.mu);
	memset(&ps.p, 0, sizeof(ps.p));
	for (auto &v : ps.iv)
		v.clear();
	for (auto &d : ps.dropped)
		d = 0;
	// time zero of the launch intervals: an event on the current device, recorded and completed now
	if (ps.base) {
		(void)hipEventDestroy(ps.base);
		ps.base = nullptr;
	}
	if (hipEventCreate(&ps.base) != hipSuccess || hipEventRecord(ps.base, nullptr) != hipSuccess || hipEventSynchronize(ps.base) != hipSuccess) {
		(void)hipGetLastError();
		if (ps.base)
			(void)hipEventDestroy(ps.base);
		ps.base = nullptr;
	}
}

extern "C" void lrzgpu_profile_get(lrzgpu_profile *out)
{
	ProfileStore &ps = ProfileStore::get();
	std::lock_guard<std::mutex> lk(ps.mu);
	*out = ps.p;
	// per kind: the wall time its launches cover (union of the intervals) and how many ran side by side at most
	for (int k = 0; k < PK_COUNT; k++) {
		std::vector<std::pair<float, int>> ev;
		ev.reserve(ps.iv[k].size() * 2);
		for (auto &iv : ps.iv[k]) {
			ev.push_back(std::make_pair(iv.first, 1));
			ev.push_back(std::make_pair(iv.second, -1));
		}
		std::sort(ev.begin(), ev.end(), [](const std::pair<float, int> &a, const std::pair<float, int> &b) {
			return a.first < b.first || (a.first == b.first && a.second < b.second);
		});
		double covered = 0;
		int depth = 0, peak = 0;
		float since = 0;
		for (auto &e : ev) {
			if (depth > 0)
				covered += e.first - since;
			since = e.first;
			depth += e.second;
			if (depth > peak)
				peak = depth;
		}
		out->union_ms[k] = covered;
		out->peak_concurrency[k] = peak;
	}
}

// the launch intervals themselves ([start, end) pairs in ms since the reset), for timelines: returns how many pairs
// exist, writes at most `cap` of them
extern "C" int lrzgpu_profile_intervals(int kind, double *out, int cap)
{
	if (kind < 0 || kind >= PK_COUNT)
		return LRZGPU_E_PARAM;
	ProfileStore &ps = ProfileStore::get();
	std::lock_guard<std::mutex> lk(ps.mu);
	const int n = (int)ps.iv[kind].size();
	for (int k = 0; k < n && k < cap; k++) {
		out[2 * k] = ps.iv[kind][(size_t)k].first;
		out[2 * k + 1] = ps.iv[kind][(size_t)k].second;
	}
	return n;
}

// ---- filters on the device: one block resident in HBM, compress direction, in place -------------------
extern "C" int lrzgpu_filter_block_dev(int filter_flag, int delta, void *d_data, int64_t n, int device)
{
	int rc = select_device(device);
	if (rc)
		return rc;
	if (n < 0 || (n && !d_data) || !filter_supported(filter_flag, delta))
		return LRZGPU_E_PARAM;
	try {
		ThreadBuffers &tb = thread_buffers();
		const size_t need = filter_scratch_bytes(filter_flag, (size_t)n);
		if (!tb.ensure(device, need))
			return LRZGPU_E_NOMEM;
		const int r = filter_block_device(filter_flag, delta, (uint8_t *)d_data, (size_t)n, tb.block.p, tb.block.cap, tb.s);
		if (r)
			return r == -1 ? LRZGPU_E_PARAM : (r == -2 ? LRZGPU_E_NOMEM : LRZGPU_E_HIP);
		return stream_wait(tb.s) == hipSuccess ? 0 : LRZGPU_E_HIP;
	} catch (...) {
		return LRZGPU_E_INTERNAL;
	}
}
