// driver.cpp -- whole-file compress driver: the rzip_fd() control flow (reference src/rzip.c:922-1264)
// over the GPU stages, with the per-block back end pipelined between the GPU and host threads.
//
//   main thread      per chunk: K1/K2 scan -> token serialisation -> K4 literal gather -> block list
//   lz4 batch        one wavefront per block of the chunk, all blocks in one launch (lz4_gate.hip)
//   GPU workers      `gpu_slots` threads, each with a HIP stream + match-finder workspace: runs the
//                    finder for the next block while earlier blocks are being parsed on the host
//   host encoders    `host_threads` threads: LZMA optimal parser + range coder (lzma_enc.cpp)
//   writer           ordered container assembly (stream_layer.cpp)
//
// Output bytes depend only on (input, control parameters), never on thread counts here.
#include <hip/hip_runtime.h>
#include <time.h>
#include <unistd.h>

#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/lrzgpu.h"
#include "common.h"
#include "lz4_gate.h"
#include "lzma_enc.h"
#include "lzma_mf.h"
#include "md5.h"
#include "profile.h"
#include "rzip_emit.h"
#include "rzip_scan.h"
#include "stream_layer.h"

using namespace lrzgpu;

extern "C" void lrzgpu_control_init(lrzgpu_control *c)
{
	memset(c, 0, sizeof(*c));
	c->compression_level = 7;                 // src/lrzip.c:1825
	c->flags = LRZGPU_FLAG_THRESHOLD;         // lz4 test on by default
	c->threshold = 100;
	long np = sysconf(_SC_NPROCESSORS_ONLN);
	c->threads = np > 0 ? (int)np : 1;        // PROCESSORS
	c->processors = c->threads;
	long pages = sysconf(_SC_PHYS_PAGES), psz = sysconf(_SC_PAGE_SIZE);
	c->ramsize = (pages > 0 && psz > 0) ? (int64_t)pages * psz : (int64_t)8 << 30; // src/lrzip.c:95-125
}

namespace {

static double now_s()
{
	struct timespec ts;
	clock_gettime(CLOCK_MONOTONIC, &ts);
	return ts.tv_sec + ts.tv_nsec * 1e-9;
}
static bool tracing() { static int t = getenv("LRZGPU_TRACE") ? 1 : 0; return t != 0; }

struct ChunkCtx {
	int index = 0;
	int64_t size = 0;
	int chunk_bytes = 0;
	uint8_t *d_stream1 = nullptr; // owned
	int64_t stream1_len = 0;
	std::vector<uint8_t> stream0;
	std::atomic<int> pending{0};
};

struct Job {
	ChunkCtx *chunk = nullptr;
	BlockRef ref{0, 0, 0};
	// lz4 gate
	int lz4_size = -1;       // LZ4 size of the whole block (single pass case), -1 = not computed
	bool lz4_ready = false;
	// finder results (host)
	std::vector<uint8_t> bytes;
	std::vector<uint8_t> counts;
	std::vector<uint32_t> pairs;
	// result
	DoneBlock done;
	bool finished = false;
};

struct Pipeline {
	lrzgpu_control *ctl;
	Sizing sz;
	int device = 0;
	int n_gpu_workers = 2, n_encoders = 1;
	int err = 0;

	std::mutex mu;
	std::condition_variable cv_jobs, cv_enc, cv_done, cv_lz4;
	std::deque<Job *> gpu_queue;  // jobs waiting for a GPU worker (file order)
	std::deque<Job *> enc_queue;  // jobs with match lists, waiting for a host encoder
	size_t enc_inflight = 0;      // queued + running encodes (bounds host memory)
	size_t enc_limit = 4;
	bool closing = false;
	double t_last_mf = 0, t_last_enc = 0, t_first_enc = 0; // trace only
	double mf_busy = 0, d2h_busy = 0, blk_busy = 0, enc_busy = 0; // summed over workers (trace only)
	std::vector<std::thread> threads;
	std::vector<std::unique_ptr<Job>> all_jobs; // file order

	void fail(int e)
	{
		std::lock_guard<std::mutex> lk(mu);
		if (!err)
			err = e;
		cv_jobs.notify_all();
		cv_enc.notify_all();
		cv_done.notify_all();
		cv_lz4.notify_all();
	}

	void finish_job(Job *j)
	{
		ChunkCtx *c = j->chunk;
		j->bytes.clear();
		j->bytes.shrink_to_fit();
		j->counts.clear();
		j->counts.shrink_to_fit();
		j->pairs.clear();
		j->pairs.shrink_to_fit();
		{
			std::lock_guard<std::mutex> lk(mu);
			j->finished = true;
			cv_done.notify_all();
		}
		if (c->pending.fetch_sub(1) == 1 && c->d_stream1) {
			(void)hipFree(c->d_stream1);
			c->d_stream1 = nullptr;
		}
	}

	void store_raw(Job *j)
	{
		j->done.c_type = CTYPE_NONE;
		j->done.payload.swap(j->bytes);
	}

	// reference lzma_compress_buf(), src/stream.c:429-494, host half
	void encoder_main()
	{
		for (;;) {
			Job *j = nullptr;
			{
				std::unique_lock<std::mutex> lk(mu);
				cv_enc.wait(lk, [&] { return !enc_queue.empty() || closing || err; });
				if (err || (enc_queue.empty() && closing))
					return;
				j = enc_queue.front();
				enc_queue.pop_front();
			}
			LzmaParams p;
			lzma_normalize(p, sz.level, sz.dict_size, 3, 0, 2, sz.level < 7 ? 32 : 64);
			MatchLists ml;
			ml.counts = j->counts.data();
			ml.pairs = j->pairs.data();
			// dlen = round_up_page(s_len * 1.02), src/stream.c:443
			size_t cap = (size_t)((double)j->ref.len * 1.02);
			cap = (cap + kPage - 1) / kPage * kPage;
			std::vector<uint8_t> dst(cap);
			size_t out_len = 0;
			int r = lzma_encode_block(p, j->bytes.data(), (size_t)j->ref.len, ml, dst.data(), cap, &out_len);
			if (r == LZ_OK && (int64_t)out_len < j->ref.len) {
				dst.resize(out_len);
				j->done.c_type = CTYPE_LZMA;
				j->done.payload.swap(dst);
			} else if (r == LZ_OK || r == LZ_ERROR_OUTPUT_EOF) {
				store_raw(j); // incompressible: stays CTYPE_NONE
			} else {
				fail(LRZGPU_E_INTERNAL);
				return;
			}
			{
				std::lock_guard<std::mutex> lk(mu);
				enc_inflight--;
				t_last_enc = now_s();
				cv_jobs.notify_all();
			}
			finish_job(j);
		}
	}

	void gpu_worker_main()
	{
		if (hipSetDevice(device) != hipSuccess) {
			fail(LRZGPU_E_HIP);
			return;
		}
		hipStream_t s;
		if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) {
			fail(LRZGPU_E_HIP);
			return;
		}
		MfWorkspace *ws = nullptr;
		uint8_t *d_stage = nullptr;
		double per_pos = 16;
		const size_t bufsize = (size_t)sz.stream_bufsize;
		auto cleanup = [&] {
			mf_workspace_destroy(ws);
			if (d_stage)
				(void)hipFree(d_stage);
			(void)hipStreamDestroy(s);
		};
		for (;;) {
			Job *j = nullptr;
			{
				std::unique_lock<std::mutex> lk(mu);
				cv_jobs.wait(lk, [&] { return err || (!gpu_queue.empty() && enc_inflight < enc_limit) || (closing && gpu_queue.empty()); });
				if (err || (gpu_queue.empty() && closing)) {
					lk.unlock();
					cleanup();
					return;
				}
				j = gpu_queue.front();
				gpu_queue.pop_front();
			}
			const double tw0 = now_s();
			const int64_t n = j->ref.len;
			j->done.streamno = j->ref.streamno;
			j->done.s_len = n;
			const bool try_backend = !sz.no_compress && n >= 64; // src/stream.c:1633
			// block bytes: device view + host copy
			const uint8_t *d_blk = nullptr;
			j->bytes.resize((size_t)n);
			if (j->ref.streamno == 0) {
				memcpy(j->bytes.data(), j->chunk->stream0.data() + j->ref.off, (size_t)n);
				if (try_backend) {
					if (!d_stage && hipMalloc(&d_stage, bufsize + 256) != hipSuccess) {
						fail(LRZGPU_E_NOMEM);
						cleanup();
						return;
					}
					if (hipMemcpyAsync(d_stage, j->bytes.data(), (size_t)n, hipMemcpyHostToDevice, s) != hipSuccess) {
						fail(LRZGPU_E_HIP);
						cleanup();
						return;
					}
					d_blk = d_stage;
				}
			} else {
				d_blk = j->chunk->d_stream1 + j->ref.off;
				if (n && hipMemcpyAsync(j->bytes.data(), d_blk, (size_t)n, hipMemcpyDeviceToHost, s) != hipSuccess) {
					fail(LRZGPU_E_HIP);
					cleanup();
					return;
				}
			}
			if (!try_backend) {
				(void)hipStreamSynchronize(s);
				store_raw(j);
				finish_job(j);
				continue;
			}
			// lz4 gate, src/stream.c:437-440.  Blocks whose gate result comes from the chunk-wide batch
			// launch do not wait for it here: the finder below runs concurrently with that launch
			// and its result is dropped if the gate says "incompressible".
			bool compressible = true;
			const bool gate_from_batch = sz.lz4_test && j->ref.streamno == 1 && n <= 100 * 1048576;
			if (sz.lz4_test && !gate_from_batch) {
				(void)hipStreamSynchronize(s);
				int pct = lrzgpu_lz4_compresses_dev(d_blk, n, sz.threshold, device);
				if (pct < 0) {
					fail(pct);
					cleanup();
					return;
				}
				compressible = pct != 0;
			}
			if (!compressible) {
				(void)hipStreamSynchronize(s);
				store_raw(j);
				finish_job(j);
				continue;
			}
			// match finder on the GPU
			LzmaParams p;
			if (lzma_normalize(p, sz.level, sz.dict_size, 3, 0, 2, sz.level < 7 ? 32 : 64) != LZ_OK) {
				fail(LRZGPU_E_PARAM);
				cleanup();
				return;
			}
			const double tw1 = now_s();
			unsigned long long total = 0;
			for (int attempt = 0;; attempt++) {
				if (!ws && mf_workspace_create(&ws, bufsize, per_pos) != 0) {
					fail(LRZGPU_E_NOMEM);
					cleanup();
					return;
				}
				int r = mf_run_device(ws, d_blk, (size_t)n, p.dict_size, (uint32_t)p.fb, p.cut(), s, &total);
				if (r == 0)
					break;
				if (r == -4 && attempt < 3) { // pool too small for this data: grow and retry
					mf_workspace_destroy(ws);
					ws = nullptr;
					per_pos *= 3;
					continue;
				}
				fail(LRZGPU_E_INTERNAL);
				cleanup();
				return;
			}
			if (gate_from_batch) {
				std::unique_lock<std::mutex> lk(mu);
				cv_lz4.wait(lk, [&] { return j->lz4_ready || err; });
				if (err) {
					lk.unlock();
					cleanup();
					return;
				}
				const int r = j->lz4_size;
				lk.unlock();
				if (lz4_compresses_decision(n, sz.threshold, [&](int, int) { return r; }) == 0) {
					(void)hipStreamSynchronize(s);
					store_raw(j);
					finish_job(j);
					continue;
				}
			}
			const double tw2 = now_s();
			j->counts.resize((size_t)n);
			j->pairs.resize((size_t)total ? (size_t)total : 1);
			if (hipMemcpyAsync(j->counts.data(), ws->counts, (size_t)n, hipMemcpyDeviceToHost, s) != hipSuccess ||
			    (total && hipMemcpyAsync(j->pairs.data(), ws->pool_out, (size_t)total * 4, hipMemcpyDeviceToHost, s) != hipSuccess) ||
			    hipStreamSynchronize(s) != hipSuccess) {
				fail(LRZGPU_E_HIP);
				cleanup();
				return;
			}
			{
				std::lock_guard<std::mutex> lk(mu);
				enc_queue.push_back(j);
				enc_inflight++;
				t_last_mf = now_s();
				blk_busy += tw1 - tw0;
				mf_busy += tw2 - tw1;
				d2h_busy += t_last_mf - tw2;
				cv_enc.notify_one();
			}
		}
	}

	void start()
	{
		for (int i = 0; i < n_gpu_workers; i++)
			threads.emplace_back([this] { gpu_worker_main(); });
		for (int i = 0; i < n_encoders; i++)
			threads.emplace_back([this] { encoder_main(); });
	}
	void stop()
	{
		{
			std::lock_guard<std::mutex> lk(mu);
			closing = true;
			cv_jobs.notify_all();
			cv_enc.notify_all();
		}
		for (auto &t : threads)
			t.join();
		threads.clear();
	}
};

struct Input {
	const uint8_t *host = nullptr; // one of host / dev
	const uint8_t *dev = nullptr;
	int fd = -1;
	int64_t n = 0;
};

int run_compress(lrzgpu_control *ctl, const Input &in, std::vector<uint8_t> *out, bool with_magic)
{
	int rc = select_device(ctl->device);
	if (rc)
		return rc;
	Pipeline P;
	P.ctl = ctl;
	P.device = ctl->device;
	rc = compute_sizing(ctl, in.n, &P.sz);
	if (rc)
		return rc;
	if (!P.sz.no_compress && P.sz.level < 5)
		return LRZGPU_E_PARAM; // levels 1-4 use the HC5 fast path: outside this library
	P.n_encoders = ctl->host_threads > 0 ? ctl->host_threads : (ctl->threads > 0 ? ctl->threads : 1);
	P.n_gpu_workers = ctl->gpu_slots > 0 ? ctl->gpu_slots : 3;
	P.enc_limit = (size_t)P.n_encoders + 2;
	ctl->stream_bufsize = P.sz.stream_bufsize;
	ctl->dictSize_used = P.sz.dict_size;
	ctl->threads_used = P.sz.threads;
	ctl->st_size = in.n;
	if (ctl->verbose)
		fprintf(stderr, "lrzgpu: threads %d bufsize %lld dict %u chunk %lld encoders %d gpu workers %d\n", P.sz.threads,
			(long long)P.sz.stream_bufsize, P.sz.dict_size, (long long)P.sz.max_chunk, P.n_encoders, P.n_gpu_workers);

	// whole-input MD5 on a side thread (the reference feeds it from cksumthread, src/rzip.c:564-584)
	uint8_t digest[16];
	std::atomic<int> md5_err{0};
	std::thread md5_thread([&] {
		Md5 m;
		if (in.host) {
			m.update(in.host, (size_t)in.n);
		} else if (in.dev) {
			if (hipSetDevice(ctl->device) != hipSuccess) {
				md5_err = LRZGPU_E_HIP;
				return;
			}
			const size_t piece = (size_t)64 << 20;
			uint8_t *stage = nullptr;
			hipStream_t s;
			if (hipHostMalloc((void **)&stage, piece, hipHostMallocDefault) != hipSuccess || hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) {
				md5_err = LRZGPU_E_NOMEM;
				return;
			}
			for (int64_t o = 0; o < in.n; o += (int64_t)piece) {
				size_t k = (size_t)(in.n - o < (int64_t)piece ? in.n - o : (int64_t)piece);
				if (hipMemcpyAsync(stage, in.dev + o, k, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) {
					md5_err = LRZGPU_E_HIP;
					break;
				}
				m.update(stage, k);
			}
			(void)hipHostFree(stage);
			(void)hipStreamDestroy(s);
		}
		m.finish(digest);
	});

	const double t0 = now_s();
	double t_scan = 0, t_gather = 0, t_lz4 = 0;
	P.start();

	std::vector<std::unique_ptr<ChunkCtx>> chunks;
	ScanWorkspace *sw = nullptr;
	int64_t victim_round = 0;
	int64_t len = in.n;
	int ret = 0;
	hipStream_t ms;
	if (hipStreamCreateWithFlags(&ms, hipStreamNonBlocking) != hipSuccess)
		ret = LRZGPU_E_HIP;
	uint8_t *d_upload = nullptr; // chunk staging when the input is on the host
	int pass = 0;
	while (!ret && (!pass || len > 0)) { // src/rzip.c:1041
		pass++;
		const int64_t offset = in.n - len;
		const int64_t chunk_size = P.sz.max_chunk < len ? P.sz.max_chunk : len;
		std::unique_ptr<ChunkCtx> cc(new ChunkCtx());
		cc->index = (int)chunks.size();
		cc->size = chunk_size;
		cc->chunk_bytes = chunk_bytes_for(chunk_size);

		const uint8_t *d_chunk = nullptr;
		bool own_chunk = false;
		if (in.dev && ((uintptr_t)(in.dev + offset) & 15) == 0 && offset + chunk_size < in.n) {
			d_chunk = in.dev + offset; // interior chunk of a resident buffer: readable past its end
		} else {
			// host input, the last chunk (needs 64 readable bytes of padding) or an unaligned view
			if (!d_upload && hipMalloc(&d_upload, (size_t)(P.sz.max_chunk < in.n ? P.sz.max_chunk : in.n) + 256) != hipSuccess) {
				ret = LRZGPU_E_NOMEM;
				break;
			}
			hipError_t e = hipSuccess;
			if (chunk_size) {
				if (in.dev)
					e = hipMemcpyAsync(d_upload, in.dev + offset, (size_t)chunk_size, hipMemcpyDeviceToDevice, ms);
				else
					e = hipMemcpyAsync(d_upload, in.host + offset, (size_t)chunk_size, hipMemcpyHostToDevice, ms);
			}
			if (e == hipSuccess)
				e = hipMemsetAsync(d_upload + chunk_size, 0, 256, ms);
			if (e != hipSuccess) {
				ret = LRZGPU_E_HIP;
				break;
			}
			d_chunk = d_upload;
			own_chunk = true;
		}
		(void)own_chunk;

		if (!sw && scan_workspace_create(&sw, P.sz.rzip_level, P.sz.max_chunk < in.n ? P.sz.max_chunk : in.n) != 0) {
			ret = LRZGPU_E_NOMEM;
			break;
		}
		ScanResult sr;
		int r = scan_chunk_device(sw, d_chunk, chunk_size, P.sz.rzip_level, &victim_round, &sr, ms);
		if (r) {
			ret = r == -4 ? LRZGPU_E_NOMEM : LRZGPU_E_INTERNAL;
			break;
		}
		t_scan = now_s();
		EmitResult er;
		emit_streams(sr.records, chunk_size, cc->chunk_bytes, sr.crc, &er);
		cc->stream0.swap(er.stream0);
		cc->stream1_len = er.stream1_len;
		if (hipMalloc(&cc->d_stream1, (size_t)er.stream1_len + 256) != hipSuccess) {
			ret = LRZGPU_E_NOMEM;
			break;
		}
		if (!er.runs.empty()) {
			CopyRun *d_runs = nullptr;
			if (hipMalloc(&d_runs, er.runs.size() * sizeof(CopyRun)) != hipSuccess) {
				ret = LRZGPU_E_NOMEM;
				break;
			}
			if (hipMemcpyAsync(d_runs, er.runs.data(), er.runs.size() * sizeof(CopyRun), hipMemcpyHostToDevice, ms) != hipSuccess) {
				(void)hipFree(d_runs);
				ret = LRZGPU_E_HIP;
				break;
			}
			EventTimer tg(ms);
			int gr = gather_runs_device(d_chunk, cc->d_stream1, d_runs, (int)er.runs.size(), er.stream1_len, ms);
			tg.stop();
			if (gr != 0 || hipMemsetAsync(cc->d_stream1 + er.stream1_len, 0, 256, ms) != hipSuccess || hipStreamSynchronize(ms) != hipSuccess) {
				(void)hipFree(d_runs);
				ret = LRZGPU_E_HIP;
				break;
			}
			(void)hipFree(d_runs);
			{
				ProfileStore &ps = ProfileStore::get();
				std::lock_guard<std::mutex> lk(ps.mu);
				ps.p.gather_ms += tg.ms();
				ps.p.gather_launches++;
				ps.p.gather_bytes += er.stream1_len;
			}
		}
		t_gather = now_s();
		// block list in flush order
		std::vector<BlockRef> refs;
		block_order(cc->stream0, cc->chunk_bytes, cc->stream1_len, P.sz.stream_bufsize, &refs);
		cc->pending = (int)refs.size();
		std::vector<Job *> new_jobs;
		{
			std::lock_guard<std::mutex> lk(P.mu);
			for (const BlockRef &br : refs) {
				std::unique_ptr<Job> j(new Job());
				j->chunk = cc.get();
				j->ref = br;
				new_jobs.push_back(j.get());
				P.all_jobs.push_back(std::move(j));
			}
		}
		// lz4 gate for all stream-1 blocks of the chunk in one launch (one wavefront per block)
		if (P.sz.lz4_test) {
			std::vector<Lz4Job> lj;
			std::vector<Job *> lz_jobs;
			for (Job *j : new_jobs)
				if (j->ref.streamno == 1 && j->ref.len >= 64 && j->ref.len <= 100 * 1048576) {
					Lz4Job q;
					q.src = cc->d_stream1 + j->ref.off;
					q.src_size = (int)j->ref.len;
					q.dst_capacity = (int)j->ref.len + 1;
					lj.push_back(q);
					lz_jobs.push_back(j);
				}
			if (!lj.empty()) {
				Lz4Job *d_jobs = nullptr;
				int *d_res = nullptr;
				if (hipMalloc(&d_jobs, lj.size() * sizeof(Lz4Job)) != hipSuccess || hipMalloc(&d_res, lj.size() * sizeof(int)) != hipSuccess) {
					ret = LRZGPU_E_NOMEM;
					break;
				}
				// enqueue the blocks for the finder first: it runs concurrently with the gate
				{
					std::lock_guard<std::mutex> lk(P.mu);
					for (Job *j : new_jobs)
						P.gpu_queue.push_back(j);
					P.cv_jobs.notify_all();
				}
				new_jobs.clear();
				std::vector<int> res(lj.size());
				if (hipMemcpyAsync(d_jobs, lj.data(), lj.size() * sizeof(Lz4Job), hipMemcpyHostToDevice, ms) != hipSuccess) {
					ret = LRZGPU_E_HIP;
					break;
				}
				EventTimer tl(ms);
				int lr = lz4_sizes_device(d_jobs, (int)lj.size(), d_res, ms);
				tl.stop();
				if (lr != 0 || hipMemcpyAsync(res.data(), d_res, lj.size() * sizeof(int), hipMemcpyDeviceToHost, ms) != hipSuccess ||
				    hipStreamSynchronize(ms) != hipSuccess) {
					ret = LRZGPU_E_HIP;
					break;
				}
				{
					ProfileStore &ps = ProfileStore::get();
					std::lock_guard<std::mutex> lk(ps.mu);
					ps.p.lz4_ms += tl.ms();
					ps.p.lz4_launches++;
					for (const Lz4Job &q : lj)
						ps.p.lz4_bytes += q.src_size;
				}
				(void)hipFree(d_jobs);
				(void)hipFree(d_res);
				std::lock_guard<std::mutex> lk(P.mu);
				for (size_t k = 0; k < lz_jobs.size(); k++) {
					lz_jobs[k]->lz4_size = res[k];
					lz_jobs[k]->lz4_ready = true;
				}
				P.cv_lz4.notify_all();
			}
		}
		if (!new_jobs.empty()) {
			std::lock_guard<std::mutex> lk(P.mu);
			for (Job *j : new_jobs)
				P.gpu_queue.push_back(j);
			P.cv_jobs.notify_all();
		}
		t_lz4 = now_s();
		chunks.push_back(std::move(cc));
		len -= chunk_size;
	}
	if (ret)
		P.fail(ret);

	// wait for every block
	{
		std::unique_lock<std::mutex> lk(P.mu);
		P.cv_done.wait(lk, [&] {
			if (P.err)
				return true;
			for (auto &j : P.all_jobs)
				if (!j->finished)
					return false;
			return true;
		});
		if (P.err && !ret)
			ret = P.err;
	}
	const double t_blocks = now_s();
	P.stop();
	md5_thread.join();
	const double t_md5 = now_s();
	if (!ret && md5_err)
		ret = md5_err;
	scan_workspace_destroy(sw);
	if (d_upload)
		(void)hipFree(d_upload);
	(void)hipStreamDestroy(ms);
	for (auto &c : chunks)
		if (c->d_stream1) {
			(void)hipFree(c->d_stream1);
			c->d_stream1 = nullptr;
		}
	if (ret)
		return ret;

	// ordered container assembly
	if (with_magic)
		out->assign(21, 0);
	size_t ji = 0;
	for (size_t ci = 0; ci < chunks.size(); ci++) {
		std::vector<DoneBlock> blocks;
		while (ji < P.all_jobs.size() && P.all_jobs[ji]->chunk == chunks[ci].get()) {
			blocks.push_back(std::move(P.all_jobs[ji]->done));
			ji++;
		}
		write_chunk(out, chunks[ci]->chunk_bytes, ci + 1 == chunks.size(), chunks[ci]->size, blocks);
	}
	out->insert(out->end(), digest, digest + 16);
	memcpy(ctl->hash_resblock, digest, 16);
	if (tracing())
		fprintf(stderr, "lrzgpu driver: scan %.2f  gather %.2f  lz4+enqueue %.2f  last finder %.2f  last encode %.2f  all blocks %.2f  md5 joined %.2f  assembled %.2f s (since start; last chunk); worker sums: block copy+gate %.2f finder %.2f lists D2H %.2f s\n",
			t_scan - t0, t_gather - t0, t_lz4 - t0, P.t_last_mf - t0, P.t_last_enc - t0, t_blocks - t0, t_md5 - t0, now_s() - t0, P.blk_busy, P.mf_busy, P.d2h_busy);
	if (with_magic) {
		uint8_t magic[21];
		write_magic(magic, P.sz, in.n);
		memcpy(out->data(), magic, 21);
	}
	{
		LzmaParams p;
		if (!P.sz.no_compress && lzma_normalize(p, P.sz.level, P.sz.dict_size, 3, 0, 2, P.sz.level < 7 ? 32 : 64) == LZ_OK)
			lzma_write_props(p, ctl->lzma_properties);
	}
	return 0;
}

int write_all(int fd, const uint8_t *p, size_t n)
{
	while (n) {
		ssize_t w = write(fd, p, n > ((size_t)1 << 30) ? ((size_t)1 << 30) : n);
		if (w <= 0)
			return LRZGPU_E_IO;
		p += w;
		n -= (size_t)w;
	}
	return 0;
}

int read_fd_all(int fd, std::vector<uint8_t> *buf)
{
	off_t end = lseek(fd, 0, SEEK_END);
	if (end < 0 || lseek(fd, 0, SEEK_SET) < 0)
		return LRZGPU_E_IO;
	buf->resize((size_t)end);
	size_t got = 0;
	while (got < (size_t)end) {
		ssize_t r = read(fd, buf->data() + got, (size_t)end - got > ((size_t)1 << 30) ? ((size_t)1 << 30) : (size_t)end - got);
		if (r <= 0)
			return LRZGPU_E_IO;
		got += (size_t)r;
	}
	return 0;
}

} // namespace

extern "C" int lrzgpu_compress_buffer(lrzgpu_control *control, const uint8_t *in, int64_t n, uint8_t **out, int64_t *out_len)
{
	if (!control || n < 0 || (!in && n))
		return LRZGPU_E_PARAM;
	Input i;
	static const uint8_t empty = 0;
	i.host = in ? in : &empty;
	i.n = n;
	std::vector<uint8_t> o;
	int r = run_compress(control, i, &o, true);
	if (r)
		return r;
	*out = (uint8_t *)malloc(o.size() ? o.size() : 1);
	if (!*out)
		return LRZGPU_E_NOMEM;
	memcpy(*out, o.data(), o.size());
	*out_len = (int64_t)o.size();
	return 0;
}

extern "C" int lrzgpu_compress_buffer_dev(lrzgpu_control *control, const void *d_in, int64_t n, uint8_t **out, int64_t *out_len)
{
	if (!control || n < 0 || (!d_in && n))
		return LRZGPU_E_PARAM;
	Input i;
	i.dev = (const uint8_t *)d_in;
	i.n = n;
	if (n == 0) {
		static const uint8_t empty = 0;
		i.dev = nullptr;
		i.host = &empty;
	}
	std::vector<uint8_t> o;
	int r = run_compress(control, i, &o, true);
	if (r)
		return r;
	*out = (uint8_t *)malloc(o.size() ? o.size() : 1);
	if (!*out)
		return LRZGPU_E_NOMEM;
	memcpy(*out, o.data(), o.size());
	*out_len = (int64_t)o.size();
	return 0;
}

extern "C" int lrzgpu_rzip_fd(lrzgpu_control *control, int fd_in, int fd_out)
{
	if (!control)
		return LRZGPU_E_PARAM;
	std::vector<uint8_t> buf;
	int r = read_fd_all(fd_in, &buf);
	if (r)
		return r;
	Input i;
	static const uint8_t empty = 0;
	i.host = buf.empty() ? &empty : buf.data();
	i.n = (int64_t)buf.size();
	std::vector<uint8_t> o;
	r = run_compress(control, i, &o, false);
	if (r)
		return r;
	return write_all(fd_out, o.data(), o.size());
}

extern "C" int lrzgpu_compress_file(lrzgpu_control *control, int fd_in, int fd_out)
{
	if (!control)
		return LRZGPU_E_PARAM;
	std::vector<uint8_t> buf;
	int r = read_fd_all(fd_in, &buf);
	if (r)
		return r;
	Input i;
	static const uint8_t empty = 0;
	i.host = buf.empty() ? &empty : buf.data();
	i.n = (int64_t)buf.size();
	std::vector<uint8_t> o;
	r = run_compress(control, i, &o, true);
	if (r)
		return r;
	return write_all(fd_out, o.data(), o.size());
}

// ---- host-only helpers (no device needed) ----------------------------------------------------

extern "C" int lrzgpu_plan(lrzgpu_control *control, int64_t st_size, int64_t *chunk_size)
{
	if (!control || st_size < 0)
		return LRZGPU_E_PARAM;
	Sizing s;
	int r = compute_sizing(control, st_size, &s);
	if (r)
		return r;
	control->stream_bufsize = s.stream_bufsize;
	control->dictSize_used = s.dict_size;
	control->threads_used = s.threads;
	control->st_size = st_size;
	if (chunk_size)
		*chunk_size = s.max_chunk < st_size ? s.max_chunk : st_size;
	return 0;
}

extern "C" int lrzgpu_container_store(lrzgpu_control *control, int64_t st_size, int n_chunks, const int64_t *chunk_sizes,
				      const uint8_t *const *stream0, const int64_t *stream0_len,
				      const uint8_t *const *stream1, const int64_t *stream1_len, const uint8_t md5[16],
				      uint8_t **out, int64_t *out_len)
{
	if (!control || n_chunks < 1)
		return LRZGPU_E_PARAM;
	Sizing s;
	int r = compute_sizing(control, st_size, &s);
	if (r)
		return r;
	std::vector<uint8_t> o(21, 0);
	for (int c = 0; c < n_chunks; c++) {
		std::vector<uint8_t> s0(stream0[c], stream0[c] + stream0_len[c]);
		const int cb = chunk_bytes_for(chunk_sizes[c]);
		std::vector<BlockRef> refs;
		block_order(s0, cb, stream1_len[c], s.stream_bufsize, &refs);
		std::vector<DoneBlock> blocks;
		for (const BlockRef &br : refs) {
			DoneBlock b;
			b.streamno = br.streamno;
			b.c_type = CTYPE_NONE;
			b.s_len = br.len;
			const uint8_t *src = br.streamno == 0 ? stream0[c] : stream1[c];
			if (br.streamno == 1 && br.off + br.len > stream1_len[c])
				return LRZGPU_E_PARAM;
			b.payload.assign(src + br.off, src + br.off + br.len);
			blocks.push_back(std::move(b));
		}
		write_chunk(&o, cb, c + 1 == n_chunks, chunk_sizes[c], blocks);
	}
	o.insert(o.end(), md5, md5 + 16);
	uint8_t magic[21];
	write_magic(magic, s, st_size);
	memcpy(o.data(), magic, 21);
	*out = (uint8_t *)malloc(o.size());
	if (!*out)
		return LRZGPU_E_NOMEM;
	memcpy(*out, o.data(), o.size());
	*out_len = (int64_t)o.size();
	return 0;
}
