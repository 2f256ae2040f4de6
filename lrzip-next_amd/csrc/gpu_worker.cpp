// gpu_worker.cpp -- the device half of a block: a GPU worker thread takes the next block (or the next stage of an early
// block), runs the match finder on it where the scan left it (stream 1, in HBM), and copies bytes and lists into the
// job's pinned host buffers for an encoder.  Reference: the match-finder thread of the reference's encoder,
// src/lzma/C/LzFindMt.c:571-729, 946-981; the early start of blocks: DESIGN.md section 5.
#include "pipeline.h"

namespace lrzgpu {

// what one worker thread owns for the whole run
struct Pipeline::GpuWorker {
	hipStream_t s = nullptr;
	MfWorkspace *ws = nullptr;
	double ws_per_pos = 0, per_pos = 16;
	DevBuf d_stage, d_scratch, d_probe;
	uint8_t *stage[2] = {nullptr, nullptr};
	size_t bufsize = 0;
	LzmaParams lp;
	bool lzma_ok = false, pack = false;
};
// the worker's state under the names the code below uses
#define GPU_WORKER_LOCALS(w) \
	hipStream_t s = w.s; \
	MfWorkspace *&ws = w.ws; \
	double &ws_per_pos = w.ws_per_pos, &per_pos = w.per_pos; \
	DevBuf &d_stage = w.d_stage, &d_scratch = w.d_scratch, &d_probe = w.d_probe; \
	uint8_t **stage = w.stage; \
	const size_t bufsize = w.bufsize; \
	const bool want_pinned = true; /* lists and block bytes land in pinned host buffers from the pool */ \
	const LzmaParams &lp = w.lp; \
	const bool lzma_ok = w.lzma_ok, pack = w.pack; \
	(void)s, (void)ws, (void)ws_per_pos, (void)per_pos, (void)d_stage, (void)d_scratch, (void)d_probe, (void)stage, (void)bufsize; \
	(void)want_pinned, (void)lp, (void)lzma_ok, (void)pack; \
	do {                   \
	} while (0)

// device -> host.  Pinned destinations take the DMA directly; pageable ones go through the worker's
// pinned staging pair (a pageable hipMemcpy is ~1 GB/s here)
int Pipeline::d2h(void *dst, bool dst_pinned, const void *d_src, size_t bytes, uint8_t *stage[2], hipStream_t s)
{
	if (!bytes)
		return 0;
	if (dst_pinned) {
		if (hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, s) != hipSuccess || stream_wait(s) != hipSuccess)
			return -1;
		return 0;
	}
	size_t off = 0, prev_off = 0, prev_len = 0;
	int k = 0;
	while (off < bytes || prev_len) {
		size_t len = 0;
		if (off < bytes) {
			len = bytes - off < STAGE_BYTES ? bytes - off : STAGE_BYTES;
			if (hipMemcpyAsync(stage[k], (const uint8_t *)d_src + off, len, hipMemcpyDeviceToHost, s) != hipSuccess)
				return -1;
		}
		if (prev_len)
			memcpy((uint8_t *)dst + prev_off, stage[k ^ 1], prev_len);
		if (stream_wait(s) != hipSuccess)
			return -1;
		prev_off = off;
		prev_len = len;
		off += len;
		k ^= 1;
	}
	return 0;
}

int Pipeline::gpu_open(GpuWorker &w)
{
	if (hipSetDevice(device) != hipSuccess || make_stream(&w.s) != hipSuccess)
		return LRZGPU_E_HIP;
	w.per_pos = mf_per_pos;
	w.bufsize = (size_t)sz.stream_bufsize;
	if (hipHostMalloc((void **)&w.stage[0], STAGE_BYTES, hipHostMallocDefault) != hipSuccess ||
	    hipHostMalloc((void **)&w.stage[1], STAGE_BYTES, hipHostMallocDefault) != hipSuccess)
		return LRZGPU_E_NOMEM;
	w.lzma_ok = lzma_normalize(w.lp, sz.level, sz.dict_size, 3, 0, 2, sz.level < 7 ? 32 : 64) == LZ_OK;
	// lists with the tail flag; one word per pair when the format allows it (lzma_mf.hip k_gather)
	w.pack = w.lzma_ok && w.lp.dict_size <= (1u << 25) && w.lp.fb <= 65;
	return 0;
}

void Pipeline::gpu_close(GpuWorker &w)
{
	WorkspacePool::get().give_mf(w.ws, w.ws_per_pos, device);
	w.ws = nullptr;
	w.d_stage.release();
	w.d_scratch.release();
	w.d_probe.release();
	for (int k = 0; k < 2; k++)
		if (w.stage[k])
			(void)hipHostFree(w.stage[k]);
	if (w.s)
		StreamPool::get().give(w.s);
}

// the finder on d_blk[0..n), a prefix of a block of block_n bytes (0: the block itself); grows the pool when the
// data needs more list entries than it holds
int Pipeline::gpu_run_finder(GpuWorker &w, const uint8_t *d_blk, size_t n, size_t block_n, unsigned long long *total)
{
	GPU_WORKER_LOCALS(w);
	for (int attempt = 0;; attempt++) {
		if (!ws) {
			ws = WorkspacePool::get().take_mf(bufsize, per_pos, device, &ws_per_pos);
			if (!ws)
				return LRZGPU_E_NOMEM;
		}
		int r = mf_run_device(ws, d_blk, n, lp.dict_size, (uint32_t)lp.fb, lp.cut(), s, total, pack ? 2 : 1, lp.fast, block_n);
		if (r == 0)
			return 0;
		if (r == -4 && attempt < 3) { // pool too small for this data: grow and retry
			mf_workspace_destroy(ws);
			ws = nullptr;
			per_pos = ws_per_pos * 3;
			continue;
		}
		if (tracing())
			fprintf(stderr, "lrzgpu finder: run on %zu bytes (block %zu) failed with %d (pool %.1f entries per byte, attempt %d)\n", n, block_n, r,
				ws_per_pos, attempt);
		return r == -4 ? LRZGPU_E_NOMEM : LRZGPU_E_INTERNAL;
	}
}

// The gate's verdict is taken for granted when a block is started early; on data it refuses (random bytes:
// BASELINE configs[4]) that would be a finder run and an encoder per block for nothing.  So the first part of
// the block goes through the gate once, as a hint: if lz4 finds nothing in it, the block waits for its
// completion like any other (the verdict that counts is the one on the whole block, as ever).
int Pipeline::gpu_early_probe(GpuWorker &w, Job *j, const uint8_t *d_blk, int64_t P)
{
	GPU_WORKER_LOCALS(w);
	// (on this worker's own stream, with its own descriptor: nothing here allocates, nothing waits actively)
	j->probed = true;
	if (!d_probe.p && !d_probe.alloc(256, device))
		return LRZGPU_E_NOMEM;
	const int in_len = (int)(P < (int64_t)256 * 1024 ? P : (int64_t)256 * 1024); // (a hint: a quarter MiB says enough, in a millisecond)
	const int below = (int)((double)in_len * ((double)sz.threshold / 100.0));
	Lz4Job q{d_blk, in_len, in_len + 1, below};
	int res = 0;
	if (hipMemcpyAsync(d_probe.p, &q, sizeof(q), hipMemcpyHostToDevice, s) != hipSuccess ||
	    lz4_sizes_device((const Lz4Job *)d_probe.p, 1, (int *)(d_probe.p + 64), s) != 0 ||
	    d2h_pageable(&res, d_probe.p + 64, sizeof(int), s) != hipSuccess)
		return LRZGPU_E_HIP;
	j->declined = !(res > 0 && res < below);
	if (tracing_events())
		fprintf(stderr, "ev %.3f %s chunk %d stream 1 off %lld len %lld\n", now_s() - g_trace_t0, j->declined ? "probe_no" : "probe_yes",
			j->chunk->index, (long long)j->ref.off, (long long)P);
	return 0;
}

// ---- one finder run of an early block: the prefix that is there, or the whole block ----
int Pipeline::gpu_early_stage(GpuWorker &w, Job *j)
{
	GPU_WORKER_LOCALS(w);
	const int64_t n = j->ref.len;
	int64_t P, from, have_bytes;
	uint64_t w_from;
	bool full;
	{
		std::lock_guard<std::mutex> lk(mu);
		full = j->full_requested;
		P = full ? n : j->stage_want;
		// a complete block that idle encoders are waiting for: a short run first, they start on its lists
		if (full && early_split && j->stage_done == 0 && !j->enc_offered && enc_waiting > 0 && n >= 8 * early_first && n >= (1 << 20)) {
			P = n / 8;
			full = false;
		}
		from = j->valid;
		w_from = j->words_at_valid;
		have_bytes = j->bytes_copied;
	}
	if (tracing_events())
		fprintf(stderr, "ev %.3f stage_start chunk %d stream 1 off %lld len %lld\n", now_s() - g_trace_t0, j->chunk->index, (long long)j->ref.off, (long long)P);
	const uint8_t *d_blk = j->chunk->stream1.p + j->ref.off;
	int64_t new_valid = from;
	uint64_t new_words_at_valid = w_from;
	bool compressible = true;
	std::unique_ptr<RawBuf<uint32_t>> regrown; // the block's list array when this run outgrows the current one
	double tw1 = now_s(), tw2 = tw1;
	const double tw0 = tw1;
	if (!full && !j->cancelled && sz.lz4_test && !j->probed) {
		const int pr = gpu_early_probe(w, j, d_blk, P);
		if (pr)
			return pr;
	}
	if (!j->cancelled && (full || (!j->declined && P - (int64_t)lp.fb - 4 > from))) {
		if (!j->bytes.p)
			j->bytes.alloc((size_t)n, want_pinned && n >= (1 << 20));
		if (!j->counts.p)
			j->counts.alloc((size_t)n, want_pinned);
		if (P > have_bytes && d2h(j->bytes.data() + have_bytes, j->bytes.pinned, d_blk + have_bytes, (size_t)(P - have_bytes), stage, s) != 0)
			return LRZGPU_E_HIP;
		have_bytes = P > have_bytes ? P : have_bytes;
		if (full && sz.lz4_test && !j->gate_needed) { // blocks outside the batched gate take the serial one
			int pct = lrzgpu_lz4_compresses_dev(d_blk, n, sz.threshold, device);
			if (pct < 0)
				return pct;
			compressible = pct != 0;
		}
		tw1 = tw2 = now_s();
		if (compressible) {
			unsigned long long total = 0;
			int fr = gpu_run_finder(w, d_blk, (size_t)P, full ? 0 : (size_t)n, &total);
			if (fr)
				return fr;
			tw2 = now_s();
			const uint64_t words = pack ? total / 2 : total;
			new_valid = full ? n : P - (int64_t)lp.fb - 4;
			if (!full) { // where the next run's lists will differ from this one's
				unsigned long long e = 0;
				if (d2h_pageable(&e, ws->offsets + new_valid, 8, s) != hipSuccess)
					return LRZGPU_E_HIP;
				new_words_at_valid = pack ? e / 2 : e;
			} else
				new_words_at_valid = words;
			uint64_t copy_from = w_from;
			uint32_t *pairs_dst = j->pairs.data();
			bool pairs_pinned = j->pairs.pinned;
			if (!j->pairs.p || words > j->pairs.n) {
				// (first run, or the block turned out denser than its first part promised.)  An encoder may be
				// reading the current array at this moment and takes the pointer under mu whenever it is told of new
				// positions (rest_cb, take_enc): the new array is filled COMPLETELY, from word 0, before it is
				// published together with `valid` in the locked section below -- never an array with holes.
				if (j->pairs.p && tracing_events())
					fprintf(stderr, "ev %.3f lists_regrown chunk %d stream 1 off %lld len %lld\n", now_s() - g_trace_t0, j->chunk->index, (long long)j->ref.off, (long long)P);
				regrown.reset(new RawBuf<uint32_t>());
				const double per = (double)words / (double)P;
				size_t cap_words = full ? (size_t)words : (size_t)(per * 1.5 * (double)n) + ((size_t)4 << 20);
				if (cap_words < words)
					cap_words = (size_t)words;
				regrown->alloc(cap_words, want_pinned);
				copy_from = 0;
				pairs_dst = regrown->data();
				pairs_pinned = regrown->pinned;
			}
			// positions below `from` are final on the host and may be being read: only what lies behind is copied
			if (d2h(j->counts.data() + from, j->counts.pinned, ws->counts + from, (size_t)(P - from), stage, s) != 0 ||
			    (words > copy_from && d2h(pairs_dst + copy_from, pairs_pinned, ws->pool_out + copy_from, (size_t)(words - copy_from) * 4, stage, s) != 0))
				return LRZGPU_E_HIP;
		}
	}
	int act = 0;
	bool drop = false;
	{
		std::lock_guard<std::mutex> lk(mu);
		n_early_stages++;
		j->bytes_copied = have_bytes;
		j->packed = pack;
		if (regrown) { // complete: now it is the block's array (the outgrown one stays alive for whoever still reads it)
			std::swap(regrown->p, j->pairs.p);
			std::swap(regrown->n, j->pairs.n);
			std::swap(regrown->cap, j->pairs.cap);
			std::swap(regrown->pinned, j->pairs.pinned);
			if (regrown->p)
				j->old_pairs.push_back(std::move(regrown));
			regrown.reset();
		}
		if (!j->cancelled) {
			if (P > j->stage_done)
				j->stage_done = P;
			if (full) {
				j->mf_done = true;
				j->compressible_mf = compressible;
				if (compressible) {
					j->valid = n;
					j->words_at_valid = new_words_at_valid;
				}
				act = route(j);
			} else if (new_valid > j->valid) {
				j->valid = new_valid;
				j->words_at_valid = new_words_at_valid;
				if (!j->enc_offered) {
					j->enc_offered = true;
					enc_queue.push_back(j);
					cv_enc.notify_one();
				}
			}
		}
		j->in_gpu = false;
		if (j->cancelled && !j->enc_offered && !j->finished)
			drop = true; // nobody on the host side has it: it ends here
		else if (!j->cancelled && !j->retiring && !j->mf_done &&
			 (j->full_requested || j->stage_want - j->stage_done >= early_step)) {
			j->queued = true;
			gpu_queue.push_back(j);
			cv_jobs.notify_all();
		}
		t_last_mf = now_s();
		blk_busy += tw1 - tw0;
		mf_busy += tw2 - tw1;
		d2h_busy += t_last_mf - tw2;
		cv_rest.notify_all();
	}
	if (act == 1 && !j->cancelled)
		store_raw(j);
	if (act == 1 || drop)
		mark_finished(j, true);
	return 0;
}

// ---- a whole block: bytes to the host, the serial gate where the batched one does not apply, finder, lists ----
int Pipeline::gpu_whole_block(GpuWorker &w, Job *j, double tw0)
{
	GPU_WORKER_LOCALS(w);
	const int64_t n = j->ref.len;
	const bool try_backend = !sz.no_compress && n >= 64 && !j->cancelled; // src/stream.c:1633
	// block bytes: device view + host copy
	const uint8_t *d_blk = nullptr;
	j->bytes.alloc((size_t)n, want_pinned && j->ref.streamno == 1 && n >= (1 << 20));
	int rc = 0;
	if (j->ref.streamno == 0) {
		memcpy(j->bytes.data(), j->chunk->stream0.data() + j->ref.off, (size_t)n);
		if (try_backend) {
			if (!d_stage.p && !d_stage.alloc(bufsize + 256, device))
				rc = LRZGPU_E_NOMEM;
			else if (hipMemcpyAsync(d_stage.p, j->bytes.data(), (size_t)n, hipMemcpyHostToDevice, s) != hipSuccess ||
				 stream_wait(s) != hipSuccess)
				rc = LRZGPU_E_HIP;
			d_blk = d_stage.p;
		}
	} else {
		uint8_t *d_lit = j->chunk->stream1.p + j->ref.off;
		d_blk = d_lit;
		// a filter over the literal block before its back end (src/stream.c:1587-1628), where the scan left it:
		// in HBM, in place (filters_gpu.hip) -- the finder, the coder's host copy and a stored block all see the
		// filtered bytes.  (A block is filtered once: a cancelled one is rebuilt by a fresh gather.)
		if (filter_flag && n && !j->cancelled) {
			const size_t need = filter_scratch_bytes(filter_flag, (size_t)n);
			if (need > d_scratch.cap && !d_scratch.alloc(filter_scratch_bytes(filter_flag, bufsize), device))
				rc = LRZGPU_E_NOMEM;
			else if (filter_block_device(filter_flag, filter_delta, d_lit, (size_t)n, d_scratch.p, d_scratch.cap, s) != 0)
				rc = LRZGPU_E_HIP;
		}
		if (!rc && n && !j->cancelled && d2h(j->bytes.data(), j->bytes.pinned, d_blk, (size_t)n, stage, s) != 0)
			rc = LRZGPU_E_HIP;
	}
	if (rc) 
		return rc;
	// blocks outside the batched gate (stream 0, > 100 MiB) take the serial gate here
	bool compressible = try_backend;
	if (try_backend && sz.lz4_test && !j->gate_needed) {
		int pct = lrzgpu_lz4_compresses_dev(d_blk, n, sz.threshold, device);
		if (pct < 0) 
			return pct;
		compressible = pct != 0;
	}
	const double tw1 = now_s();
	double tw2 = tw1;
	if (compressible && !sz.zstd) {
		// match finder on the GPU (runs concurrently with the gate launch of this block)
		unsigned long long total = 0;
		int fr = gpu_run_finder(w, d_blk, (size_t)n, 0, &total);
		if (fr) 
			return fr;
		tw2 = now_s();
		const size_t words = pack ? (size_t)(total / 2) : (size_t)total;
		j->counts.alloc((size_t)n, want_pinned);
		j->pairs.alloc(words, want_pinned);
		j->packed = pack;
		if (d2h(j->counts.data(), j->counts.pinned, ws->counts, (size_t)n, stage, s) != 0 ||
		    (words && d2h(j->pairs.data(), j->pairs.pinned, ws->pool_out, words * 4, stage, s) != 0)) 
			return LRZGPU_E_HIP;
	}
	TRACE_EVENT("gpu_end", j);
	int act;
	{
		std::lock_guard<std::mutex> lk(mu);
		j->mf_done = true;
		j->compressible_mf = compressible;
		j->in_gpu = false;
		act = route(j);
		t_last_mf = now_s();
		blk_busy += tw1 - tw0;
		mf_busy += tw2 - tw1;
		d2h_busy += t_last_mf - tw2;
	}
	if (act == 1) {
		if (!j->cancelled)
			store_raw(j);
		mark_finished(j, true);
	}
	return 0;
}

void Pipeline::gpu_worker_main()
{
	GpuWorker w;
	int rc = gpu_open(w);
	while (!rc) {
		Job *j = nullptr;
		{
			std::unique_lock<std::mutex> lk(mu);
			cv_jobs.wait(lk, [&] { return err || (closing && gpu_queue.empty()) || (j = take_gpu()) != nullptr; });
			if (!j)
				break;
			j->queued = false;
			j->in_gpu = true;
			if (!j->held_slot) {
				j->held_slot = true;
				held++; // released in finish_locked
			}
		}
		const double tw0 = now_s();
		TRACE_EVENT("gpu_start", j);
		j->done.streamno = j->ref.streamno;
		j->done.s_len = j->ref.len;
		if (!sz.no_compress && j->ref.len >= 64 && !j->cancelled && !sz.zstd && !w.lzma_ok) {
			rc = LRZGPU_E_PARAM;
			break;
		}
		if (!j->early) {
			rc = gpu_whole_block(w, j, tw0);
			continue;
		}
		// whichever way a run ends, the job must not stay marked "in a finder run": its encoder waits for that mark to
		// clear before it lets go of the block's buffers (retire), and would wait for ever
		auto left_the_gpu = [&] {
			std::lock_guard<std::mutex> lk(mu);
			j->in_gpu = false;
			cv_rest.notify_all();
		};
		try {
			rc = gpu_early_stage(w, j);
		} catch (...) {
			left_the_gpu();
			gpu_close(w);
			throw;
		}
		TRACE_EVENT("gpu_end", j);
		if (rc)
			left_the_gpu();
	}
	if (rc)
		fail(rc);
	gpu_close(w);
}

} // namespace lrzgpu
