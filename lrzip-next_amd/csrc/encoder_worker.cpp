// encoder_worker.cpp -- the host half of a block: an encoder thread takes the next block that has match lists (or an early
// block whose lists are still arriving), parses and range-codes it (lzma_parser.cpp) or hands it to zstd, and leaves
// the result with the job.  Reference: lzma_compress_buf() / zstd_compress_buf(), src/stream.c:167-230, 429-494.
#include "pipeline.h"

namespace lrzgpu {

// StagedLists::rest: blocks until the finder has covered more of the block (or all of it and the gate agreed)
const MatchLists *Pipeline::rest_cb(void *ctx, size_t *valid)
{
	RestCtx *r = (RestCtx *)ctx;
	Pipeline *P = r->P;
	Job *j = r->j;
	const double t0 = now_s();
	std::unique_lock<std::mutex> lk(P->mu);
	P->cv_rest.wait(lk, [&] { return P->err || j->cancelled || j->refused || j->full_ready || j->valid > r->seen; });
	r->waited += now_s() - t0;
	if (P->err || j->cancelled || j->refused)
		return nullptr;
	r->seen = j->full_ready ? j->ref.len : j->valid;
	if (tracing_events())
		fprintf(stderr, "ev %.3f rest chunk %d stream 1 off %lld len %lld waited %.3f\n", now_s() - g_trace_t0, j->chunk->index, (long long)j->ref.off, (long long)r->seen, now_s() - t0);
	r->ml.counts = j->counts.data();
	r->ml.pairs = j->pairs.data(); // (may have moved: an outgrown array stays alive in old_pairs)
	*valid = (size_t)r->seen;
	return &r->ml;
}

// the encoder is through with an early block: no further finder run on it, and none still running
void Pipeline::retire(Job *j)
{
	std::unique_lock<std::mutex> lk(mu);
	j->retiring = true;
	if (j->queued) {
		for (size_t i = 0; i < gpu_queue.size(); i++)
			if (gpu_queue[i] == j) {
				gpu_queue.erase(gpu_queue.begin() + (long)i);
				break;
			}
		j->queued = false;
	}
	cv_rest.wait(lk, [&] { return !j->in_gpu; });
}


// reference lzma_compress_buf(), src/stream.c:429-494, host half
void Pipeline::encoder_main()
{
	for (;;) {
		Job *j = nullptr;
		const double tw0 = now_s();
		bool staged = false;
		RestCtx rcx{this, nullptr, 0, MatchLists(), 0};
		{
			std::unique_lock<std::mutex> lk(mu);
			enc_waiting++;
			cv_enc.wait(lk, [&] { return !enc_queue.empty() || closing || err; });
			enc_waiting--;
			if (err || (enc_queue.empty() && closing))
				return;
			j = take_enc();
			if (j->early) {
				if (!j->with_encoder)
					early_unclaimed--;
				j->with_encoder = true;
				staged = true;
				rcx.j = j;
				rcx.seen = j->full_ready ? j->ref.len : j->valid;
				rcx.ml.counts = j->counts.data();
				rcx.ml.pairs = j->pairs.data();
				rcx.ml.packed = j->packed;
				rcx.ml.tail_flags = true;
			}
			if (t_first_enc == 0)
				t_first_enc = now_s();
		}
		const double te0 = now_s();
		TRACE_EVENT("enc_start", j);
		if (!j->cancelled && sz.zstd) {
			// zstd_compress_buf(), src/stream.c:167-230: dlen = round_up_page(s_len); "does not fit" and
			// "not smaller" both leave the block stored
			const ZstdLib &z = ZstdLib::get();
			size_t cap = ((size_t)j->ref.len + kPage - 1) / kPage * kPage;
			RawBuf<uint8_t> dst;
			dst.alloc(cap);
			const size_t r = z.compress(dst.data(), cap, j->bytes.data(), (size_t)j->ref.len, sz.zstd_level);
			if (z.is_error(r)) {
				if ((size_t)0 - r != 70) { // ZSTD_error_dstSize_tooSmall = incompressible
					fail(LRZGPU_E_INTERNAL);
					return;
				}
				store_raw(j);
			} else if ((int64_t)r >= j->ref.len) {
				store_raw(j);
			} else {
				j->done.c_type = CTYPE_ZSTD;
				j->done.payload.assign(dst.data(), dst.data() + r);
			}
		} else if (!j->cancelled) {
			LzmaParams p;
			lzma_normalize(p, sz.level, sz.dict_size, 3, 0, 2, sz.level < 7 ? 32 : 64);
			// dlen = round_up_page(s_len * 1.02), src/stream.c:443
			size_t cap = (size_t)((double)j->ref.len * 1.02);
			cap = (cap + kPage - 1) / kPage * kPage;
			RawBuf<uint8_t> dst;
			dst.alloc(cap);
			size_t out_len = 0;
			int r;
			if (staged) {
				// the lists arrive while the parse runs (lzma_enc.h StagedLists); the gate's verdict was taken
				// for granted: a refusal withdraws the block (rest_cb returns nullptr) and it is stored
				StagedLists sl;
				sl.early = rcx.ml;
				sl.early_positions = (size_t)rcx.seen;
				sl.rest = &Pipeline::rest_cb;
				sl.ctx = &rcx;
				r = lzma_encode_block_staged(p, j->bytes.data(), (size_t)j->ref.len, sl, dst.data(), cap, &out_len);
				// whatever the parser said, the verdict on the block needs all of it (an overflow of dst can end the
				// parse before the block is complete; a stored block needs every byte on the host)
				std::unique_lock<std::mutex> lk(mu);
				const double t0 = now_s();
				cv_rest.wait(lk, [&] { return err || j->cancelled || j->refused || j->full_ready; });
				rcx.waited += now_s() - t0;
				if (err)
					return;
				if (j->cancelled || j->refused)
					r = j->refused ? LZ_ERROR_OUTPUT_EOF : LZ_OK; // (stored / dropped below)
			} else {
				MatchLists ml;
				ml.counts = j->counts.data();
				ml.pairs = j->pairs.data();
				ml.packed = j->packed;
				ml.tail_flags = true;
				r = lzma_encode_block(p, j->bytes.data(), (size_t)j->ref.len, ml, dst.data(), cap, &out_len);
			}
			if (j->cancelled) {
				// withdrawn: nothing of it is used
			} else if (r == LZ_OK && (int64_t)out_len < j->ref.len) {
				j->done.c_type = CTYPE_LZMA;
				j->done.payload.assign(dst.data(), dst.data() + out_len);
			} else if (r == LZ_OK || r == LZ_ERROR_OUTPUT_EOF) {
				store_raw(j); // incompressible: stays CTYPE_NONE
			} else {
				fail(LRZGPU_E_INTERNAL);
				return;
			}
		}
		if (staged)
			retire(j);
		{
			std::lock_guard<std::mutex> lk(mu);
			t_last_enc = now_s();
			enc_busy += t_last_enc - te0 - rcx.waited;
			enc_wait += te0 - tw0 + rcx.waited;
			rest_wait += rcx.waited;
		}
		TRACE_EVENT("enc_end", j);
		mark_finished(j, true);
	}
}


} // namespace lrzgpu
