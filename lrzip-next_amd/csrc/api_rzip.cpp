// api_rzip.cpp -- C ABI of the rzip stage: hash_search() over one chunk (src/rzip.c:586-762),
// token/literal emission of put_match / put_literal (src/rzip.c:208-265).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/lrzgpu.h"
#include "common.h"
#include "pools.h"
#include "rzip_emit.h"
#include "rzip_scan.h"

using namespace lrzgpu;

extern "C" void lrzgpu_hash_index(uint64_t out[256]) { hash_index_table(out); }

namespace lrzgpu {

static inline void put_le(std::vector<uint8_t> &v, uint64_t x, int n)
{
	for (int i = 0; i < n; i++)
		v.push_back((uint8_t)(x >> (8 * i)));
}

// put_literal(), src/rzip.c:248-265
static void emit_literal(EmitResult &e, int64_t last, int64_t p)
{
	if (p > last) {
		if (!e.runs.empty() && e.runs.back().src_off + e.runs.back().len == last)
			e.runs.back().len += p - last;
		else {
			CopyRun r;
			r.src_off = last;
			r.dst_off = e.stream1_len;
			r.len = p - last;
			e.runs.push_back(r);
		}
	}
	do {
		int64_t len = p - last;
		if (len > 0xFFFF)
			len = 0xFFFF;
		e.literals++;
		e.literal_bytes += len;
		e.stream0.push_back(0);
		put_le(e.stream0, (uint64_t)len, 2);
		e.stream1_len += len;
		last += len;
	} while (p > last);
}

// put_match(), src/rzip.c:208-226
static void emit_match(EmitResult &e, int64_t p, int64_t offset, int64_t len, int chunk_bytes)
{
	do {
		int64_t n = len > 0xFFFF ? 0xFFFF : len;
		e.stream0.push_back(1);
		put_le(e.stream0, (uint64_t)n, 2);
		put_le(e.stream0, (uint64_t)(p - offset), chunk_bytes);
		e.matches++;
		e.match_bytes += n;
		len -= n;
		p += n;
		offset += n;
	} while (len);
}

void emit_streams(const std::vector<MatchRec> &recs, int64_t chunk_size, int chunk_bytes, uint32_t crc, EmitResult *out)
{
	EmitResult &e = *out;
	e = EmitResult();
	int64_t last = 0;
	for (const MatchRec &r : recs) {
		if (last < r.p)
			emit_literal(e, last, r.p);
		emit_match(e, r.p, r.ofs, r.len, chunk_bytes);
		last = r.p + r.len;
	}
	if (last < chunk_size)
		emit_literal(e, last, chunk_size);
	// terminator put_literal(0,0) and the CRC, most significant byte first (src/rzip.c:757-760)
	emit_literal(e, 0, 0);
	e.stream0.push_back((uint8_t)(crc >> 24));
	e.stream0.push_back((uint8_t)(crc >> 16));
	e.stream0.push_back((uint8_t)(crc >> 8));
	e.stream0.push_back((uint8_t)crc);
}

} // namespace lrzgpu

// K1 of the scan on its own: the candidate positions of d_chunk[first .. chunk_size - 31] whose 31-byte XOR tag has all
// bits of min_mask set (src/rzip.c:385-416, 654-659), as the resolver would be handed them.
extern "C" int lrzgpu_tag_candidates_dev(const void *d_chunk, int64_t chunk_size, int64_t first, uint64_t min_mask, int reps,
					 int64_t *count, uint64_t *checksum, double *ms_per_pass, int only_tags, int device)
{
	int rc = select_device(device);
	if (rc)
		return rc;
	if (chunk_size < 0 || first < 0 || !count || !checksum || ((uintptr_t)d_chunk & 15) != 0)
		return LRZGPU_E_PARAM;
	*count = 0;
	*checksum = 0;
	if (ms_per_pass)
		*ms_per_pass = 0;
	const int64_t end = chunk_size - 31;
	if (end < first)
		return 0;
	ScanWorkspace *w = nullptr;
	if (scan_workspace_create(&w, 1, chunk_size) != 0) {
		scan_workspace_destroy(w);
		return LRZGPU_E_NOMEM;
	}
	unsigned long long *d_out = nullptr, h_out[2] = {0, 0};
	rc = LRZGPU_E_HIP;
	// (a stream of its own: on the null stream every launch would first be ordered behind the blocking scan streams a
	// process that has compressed before keeps parked -- 0.6 instead of 1.0 TB/s in bench.py's figure)
	hipStream_t ks = pooled_stream(device);
	if (ks && hipMalloc(&d_out, 16) == hipSuccess && tag_candidates_device(w, (const uint8_t *)d_chunk, first, end, min_mask, reps, d_out, ms_per_pass, ks, only_tags != 0) == 0 &&
	    hipMemcpyAsync(h_out, d_out, 16, hipMemcpyDeviceToHost, ks) == hipSuccess && stream_wait(ks) == hipSuccess) {
		*count = (int64_t)h_out[0];
		*checksum = (uint64_t)h_out[1];
		rc = 0;
	}
	if (ks) {
		(void)stream_wait(ks);
		StreamPool::get().give(ks);
	}
	if (d_out)
		(void)hipFree(d_out);
	scan_workspace_destroy(w);
	return rc;
}

extern "C" int lrzgpu_hash_search_dev(const void *d_chunk, int64_t chunk_size, int rzip_level, int chunk_bytes,
				      int64_t *victim_round, uint8_t **stream0, int64_t *stream0_len,
				      void *d_stream1, int64_t *stream1_len, uint32_t *crc32, lrzgpu_scan_stats *stats,
				      int device)
{
	int rc = select_device(device);
	if (rc)
		return rc;
	if (chunk_size < 0 || rzip_level < 0 || rzip_level > 9 || chunk_bytes < 1 || chunk_bytes > 8)
		return LRZGPU_E_PARAM;
	if (((uintptr_t)d_chunk & 15) != 0)
		return LRZGPU_E_PARAM; // chunk base must be 16-byte aligned (and padded by 64 readable bytes)
	ScanWorkspace *w = nullptr;
	if (scan_workspace_create(&w, rzip_level, chunk_size) != 0) {
		scan_workspace_destroy(w);
		return LRZGPU_E_NOMEM;
	}
	ScanResult res;
	int64_t vr = victim_round ? *victim_round : 0;
	int r = scan_chunk_device(w, (const uint8_t *)d_chunk, chunk_size, rzip_level, &vr, &res, 0);
	if (r != 0) {
		scan_workspace_destroy(w);
		return r == -4 ? LRZGPU_E_NOMEM : LRZGPU_E_INTERNAL;
	}
	if (victim_round)
		*victim_round = vr;
	EmitResult e;
	emit_streams(res.records, chunk_size, chunk_bytes, res.crc, &e);
	// stream 1: gather the literal runs on the device
	if (d_stream1 && !e.runs.empty()) {
		CopyRun *d_runs = nullptr;
		if (hipMalloc(&d_runs, e.runs.size() * sizeof(CopyRun)) != hipSuccess ||
		    hipMemcpy(d_runs, e.runs.data(), e.runs.size() * sizeof(CopyRun), hipMemcpyHostToDevice) != hipSuccess ||
		    gather_runs_device((const uint8_t *)d_chunk, (uint8_t *)d_stream1, d_runs, (int)e.runs.size(), 0, e.stream1_len, 0) != 0 ||
		    hipDeviceSynchronize() != hipSuccess) {
			if (d_runs)
				(void)hipFree(d_runs);
			scan_workspace_destroy(w);
			return LRZGPU_E_HIP;
		}
		(void)hipFree(d_runs);
	}
	*stream0_len = (int64_t)e.stream0.size();
	*stream0 = (uint8_t *)malloc(e.stream0.size() ? e.stream0.size() : 1);
	if (!*stream0) {
		scan_workspace_destroy(w);
		return LRZGPU_E_NOMEM;
	}
	memcpy(*stream0, e.stream0.data(), e.stream0.size());
	*stream1_len = e.stream1_len;
	if (crc32)
		*crc32 = res.crc;
	if (stats) {
		const ScanState &f = res.final_state;
		stats->matches = e.matches;
		stats->match_bytes = e.match_bytes;
		stats->literals = e.literals;
		stats->literal_bytes = e.literal_bytes;
		stats->inserts = f.inserts;
		stats->lookups = f.lookups;
		stats->tag_hits = f.tag_hits;
		stats->tag_misses = f.tag_misses;
		stats->hash_count = f.hash_count;
		stats->tag_clean_ptr = f.clean_ptr;
		stats->minimum_tag_mask = f.min_mask;
		stats->tag_mask = f.tag_mask;
	}
	scan_workspace_destroy(w);
	return 0;
}

extern "C" int lrzgpu_hash_search(const uint8_t *chunk, int64_t chunk_size, int rzip_level, int chunk_bytes,
				  int64_t *victim_round, uint8_t **stream0, int64_t *stream0_len,
				  uint8_t *stream1, int64_t *stream1_len, uint32_t *crc32, lrzgpu_scan_stats *stats,
				  int device)
{
	int rc = select_device(device);
	if (rc)
		return rc;
	if (chunk_size < 0)
		return LRZGPU_E_PARAM;
	uint8_t *d_chunk = nullptr, *d_s1 = nullptr;
	if (hipMalloc(&d_chunk, (size_t)chunk_size + 256) != hipSuccess)
		return LRZGPU_E_NOMEM;
	if (hipMalloc(&d_s1, (size_t)chunk_size + 256) != hipSuccess) {
		(void)hipFree(d_chunk);
		return LRZGPU_E_NOMEM;
	}
	int r = LRZGPU_E_HIP;
	if (hipMemset(d_chunk + chunk_size, 0, 256) == hipSuccess &&
	    (chunk_size == 0 || hipMemcpy(d_chunk, chunk, (size_t)chunk_size, hipMemcpyHostToDevice) == hipSuccess)) {
		r = lrzgpu_hash_search_dev(d_chunk, chunk_size, rzip_level, chunk_bytes, victim_round, stream0, stream0_len, d_s1,
					   stream1_len, crc32, stats, device);
		if (r == 0 && *stream1_len > 0 && hipMemcpy(stream1, d_s1, (size_t)*stream1_len, hipMemcpyDeviceToHost) != hipSuccess)
			r = LRZGPU_E_HIP;
	}
	(void)hipFree(d_chunk);
	(void)hipFree(d_s1);
	return r;
}
