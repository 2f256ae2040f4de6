// driver.cpp -- whole-file compress driver: the rzip_fd() control flow (reference src/rzip.c:922-1264)
// over the GPU stages, with the per-block back end pipelined between the GPU and host threads.
//
//   main thread      per chunk: K1/K2 scan segments.  After every segment the literal bytes that are
//                    already decided are gathered (K4) and every stream-1 block they complete is
//                    handed to the back end AT ONCE, so the back end runs under the scan -- the
//                    reference overlaps the same way through flush_buffer() (src/rzip.c:229-246).
//                    "Decided" = behind every emitted match, or more than SPEC_MARGIN behind the
//                    scan position; a later match reaching back over such bytes is detected and
//                    the chunk's early blocks are then thrown away and redone (not observed on the
//                    bench workloads; tests force it with LRZGPU_SPEC_MARGIN / LRZGPU_SEG_BYTES).
//   lz4 gate         one wavefront per block, launched per group of new blocks on its own stream
//   GPU workers      `gpu_slots` threads, each with a HIP stream + match-finder workspace + pinned
//                    staging: finder for block k+1 while block k is parsed on the host
//   host encoders    `host_threads` threads (default: the CPUs the process may use): LZMA parser +
//                    range coder (lzma_enc.cpp); all stream waits are blocking-sync events
//   writer           ordered container assembly (stream_layer.cpp); the file order of the blocks is
//                    block_order()'s replay of the reference flushes, whatever order they finished in
//
// Output bytes depend only on (input, control parameters), never on thread counts or timing here.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <sched.h>

#include <cerrno>
#include <time.h>
#include <unistd.h>

#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
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

// literal bytes this far behind the scan count as decided (LRZGPU_SPEC_MARGIN overrides: test hook
// for the roll-back path -- with 0 every match that extends backwards over a segment boundary violates)
static int64_t spec_margin()
{
	const char *e = getenv("LRZGPU_SPEC_MARGIN"); // read per call: tests flip it inside one process
	return e ? (int64_t)atoll(e) : (int64_t)2 << 20;
}
constexpr size_t STAGE_BYTES = (size_t)32 << 20;  // pinned D2H staging piece (two per GPU worker)

double now_s()
{
	struct timespec ts;
	clock_gettime(CLOCK_MONOTONIC, &ts);
	return ts.tv_sec + ts.tv_nsec * 1e-9;
}
bool tracing()
{
	static int t = getenv("LRZGPU_TRACE") ? 1 : 0;
	return t != 0;
}

// The resolver is one latency-bound wavefront; finder / gate kernels of the blocks emitted early would
// otherwise share its CU (issue slots, L1, LDS).  The scan stream therefore owns a small set of CUs
// and every back-end stream gets the complement (hipExtStreamCreateWithCUMask).
int scan_cu_words(uint32_t scan_mask[8], uint32_t rest_mask[8])
{
	int ncu = 256;
	hipDeviceProp_t prop;
	int dev = 0;
	if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
		ncu = prop.multiProcessorCount;
	if (ncu > 256)
		ncu = 256;
	const char *e = getenv("LRZGPU_SCAN_CUS");
	int k = e ? atoi(e) : 0; // off by default: measured no gain, and masked streams are blocking streams
	if (k <= 0 || k >= ncu)
		return 0; // masking disabled
	for (int w = 0; w < 8; w++)
		scan_mask[w] = rest_mask[w] = 0;
	for (int c = 0; c < ncu; c++) {
		if (c < k)
			scan_mask[c >> 5] |= 1u << (c & 31);
		else
			rest_mask[c >> 5] |= 1u << (c & 31);
	}
	return (ncu + 31) / 32;
}
hipError_t make_stream(hipStream_t *s, int words, const uint32_t *mask, bool high_priority = false)
{
	if (words > 0)
		return hipExtStreamCreateWithCUMask(s, (uint32_t)words, mask);
	if (high_priority) {
		// the scan stream must never queue behind a multi-second gate/finder kernel: streams share a
		// small pool of hardware queues (GPU_MAX_HW_QUEUES), priority streams get their own
		int lo = 0, hi = 0;
		if (hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess && hi != lo)
			return hipStreamCreateWithPriority(s, hipStreamNonBlocking, hi);
	}
	return hipStreamCreateWithFlags(s, hipStreamNonBlocking);
}

struct ChunkCtx {
	int index = 0;
	int64_t size = 0;
	int chunk_bytes = 0;
	uint8_t *d_stream1 = nullptr; // owned, chunk_size + 256 bytes
	int64_t stream1_len = 0;
	std::vector<uint8_t> stream0;
};

// --zstd back end: the system libzstd, bound at run time like the reference links it
// (src/stream.c:167-230 zstd_compress_buf; bit-exactness holds against the same libzstd build)
struct ZstdLib {
	size_t (*compress)(void *, size_t, const void *, size_t, int) = nullptr;
	unsigned (*is_error)(size_t) = nullptr;
	bool ok = false;
	static const ZstdLib &get()
	{
		static const ZstdLib z = [] {
			ZstdLib l;
			void *h = dlopen("libzstd.so.1", RTLD_NOW | RTLD_GLOBAL);
			if (!h)
				h = dlopen("libzstd.so", RTLD_NOW | RTLD_GLOBAL);
			if (h) {
				l.compress = (size_t(*)(void *, size_t, const void *, size_t, int))dlsym(h, "ZSTD_compress");
				l.is_error = (unsigned (*)(size_t))dlsym(h, "ZSTD_isError");
				l.ok = l.compress && l.is_error;
			}
			return l;
		}();
		return z;
	}
};

// Recycled host buffers for block copies and match lists.  A 16 MiB block has ~450 MB of lists;
// taking that from malloc() for every block means a fresh mmap, a page fault per 4 KiB and an
// munmap -- seconds of kernel time per GiB of input that the encoders would rather have.
struct HostPool {
	std::mutex mu;
	std::vector<std::pair<size_t, void *>> idle; // (capacity, pointer)
	size_t idle_bytes = 0;
	static HostPool &get()
	{
		static HostPool p;
		return p;
	}
	static size_t idle_limit() // keep at most 1/8 of physical memory (and at most 48 GiB) parked
	{
		static const size_t lim = [] {
			const long pages = sysconf(_SC_PHYS_PAGES), psz = sysconf(_SC_PAGE_SIZE);
			size_t phys = pages > 0 && psz > 0 ? (size_t)pages * (size_t)psz : (size_t)64 << 30;
			size_t l = phys / 8;
			return l > ((size_t)48 << 30) ? (size_t)48 << 30 : l;
		}();
		return lim;
	}
	void *take(size_t bytes, size_t *cap)
	{
		{
			std::lock_guard<std::mutex> lk(mu);
			size_t best = idle.size();
			for (size_t i = 0; i < idle.size(); i++)
				if (idle[i].first >= bytes && (best == idle.size() || idle[i].first < idle[best].first))
					best = i;
			if (best != idle.size() && idle[best].first <= 2 * bytes + ((size_t)64 << 20)) {
				void *p = idle[best].second;
				*cap = idle[best].first;
				idle_bytes -= idle[best].first;
				idle[best] = idle.back();
				idle.pop_back();
				return p;
			}
		}
		const size_t round = (size_t)8 << 20;
		*cap = (bytes + round - 1) / round * round;
		if (*cap == 0)
			*cap = round;
		return malloc(*cap);
	}
	void give(void *p, size_t cap)
	{
		if (!p)
			return;
		std::lock_guard<std::mutex> lk(mu);
		if (idle.size() >= 96 || idle_bytes + cap > idle_limit()) {
			free(p);
			return;
		}
		idle.emplace_back(cap, p);
		idle_bytes += cap;
	}
	~HostPool()
	{
		for (auto &e : idle)
			free(e.second);
	}
};

template <typename T> struct RawBuf { // uninitialised host buffer (std::vector would zero-fill), recycled
	T *p = nullptr;
	size_t n = 0, cap = 0;
	RawBuf() = default;
	RawBuf(const RawBuf &) = delete;
	RawBuf &operator=(const RawBuf &) = delete;
	void alloc(size_t k)
	{
		release();
		p = (T *)HostPool::get().take((k ? k : 1) * sizeof(T), &cap);
		if (!p)
			throw std::bad_alloc();
		n = k;
	}
	void release()
	{
		if (p)
			HostPool::get().give(p, cap);
		p = nullptr;
		n = cap = 0;
	}
	~RawBuf() { release(); }
	T *data() { return p; }
};

struct Job {
	ChunkCtx *chunk = nullptr;
	BlockRef ref{0, 0, 0};
	// state, guarded by Pipeline::mu
	bool gate_needed = false; // lz4 result comes from a batch launch
	bool lz4_ready = false;
	int lz4_size = -1;
	bool mf_done = false;
	bool compressible_mf = false; // finder ran and produced lists
	bool dispatched = false;
	bool cancelled = false;
	bool finished = false;
	// data
	RawBuf<uint8_t> bytes;
	RawBuf<uint8_t> counts;
	RawBuf<uint32_t> pairs;
	bool packed = false;
	DoneBlock done;
};

struct Lz4Batch {
	hipEvent_t ev = nullptr;
	Lz4Job *d_jobs = nullptr;
	int *d_res = nullptr;
	std::vector<Job *> jobs;
	EventTimer *timer = nullptr;
	int64_t bytes = 0;
};

struct Pipeline {
	lrzgpu_control *ctl;
	Sizing sz;
	int device = 0;
	int mask_words = 0;
	uint32_t scan_mask[8], rest_mask[8];
	int n_gpu_workers = 2, n_encoders = 1;
	int err = 0;

	std::mutex mu;
	std::condition_variable cv_jobs, cv_enc, cv_done;
	std::deque<Job *> gpu_queue; // blocks waiting for a GPU worker
	std::deque<Job *> enc_queue; // blocks with match lists and a positive gate, waiting for a host encoder
	size_t held = 0;             // blocks holding host match lists (bounds host memory)
	size_t held_limit = 4;
	bool closing = false;
	double t_last_mf = 0, t_last_enc = 0;
	double mf_busy = 0, d2h_busy = 0, blk_busy = 0, enc_busy = 0, enc_wait = 0;
	std::vector<std::thread> threads;

	void fail(int e)
	{
		std::lock_guard<std::mutex> lk(mu);
		if (!err)
			err = e;
		cv_jobs.notify_all();
		cv_enc.notify_all();
		cv_done.notify_all();
	}

	void mark_finished(Job *j, bool held_lists)
	{
		j->bytes.release();
		j->counts.release();
		j->pairs.release();
		std::lock_guard<std::mutex> lk(mu);
		if (held_lists) {
			held--;
			cv_jobs.notify_all();
		}
		j->finished = true;
		cv_done.notify_all();
	}

	void store_raw(Job *j)
	{
		j->done.c_type = CTYPE_NONE;
		j->done.payload.assign(j->bytes.data(), j->bytes.data() + j->ref.len);
	}

	// Called with mu held whenever the finder result or the gate result of a block arrives: once both
	// are there the block either goes to the encoders or is stored.  Returns 1 if the caller must
	// finish the block as stored (outside the lock).
	int route(Job *j)
	{
		if (j->dispatched || !j->mf_done || (j->gate_needed && !j->lz4_ready))
			return 0;
		j->dispatched = true;
		bool compressible = j->compressible_mf && !j->cancelled;
		if (compressible && j->gate_needed)
			compressible = lz4_compresses_decision(j->ref.len, sz.threshold, [&](int, int) { return j->lz4_size; }) != 0;
		if (compressible) {
			enc_queue.push_back(j);
			cv_enc.notify_one();
			return 0;
		}
		return 1;
	}

	// reference lzma_compress_buf(), src/stream.c:429-494, host half
	void encoder_main()
	{
		for (;;) {
			Job *j = nullptr;
			const double tw0 = now_s();
			{
				std::unique_lock<std::mutex> lk(mu);
				cv_enc.wait(lk, [&] { return !enc_queue.empty() || closing || err; });
				if (err || (enc_queue.empty() && closing))
					return;
				j = enc_queue.front();
				enc_queue.pop_front();
			}
			const double te0 = now_s();
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
				MatchLists ml;
				ml.counts = j->counts.data();
				ml.pairs = j->pairs.data();
				ml.packed = j->packed;
				// dlen = round_up_page(s_len * 1.02), src/stream.c:443
				size_t cap = (size_t)((double)j->ref.len * 1.02);
				cap = (cap + kPage - 1) / kPage * kPage;
				RawBuf<uint8_t> dst;
				dst.alloc(cap);
				size_t out_len = 0;
				int r = lzma_encode_block(p, j->bytes.data(), (size_t)j->ref.len, ml, dst.data(), cap, &out_len);
				if (r == LZ_OK && (int64_t)out_len < j->ref.len) {
					j->done.c_type = CTYPE_LZMA;
					j->done.payload.assign(dst.data(), dst.data() + out_len);
				} else if (r == LZ_OK || r == LZ_ERROR_OUTPUT_EOF) {
					store_raw(j); // incompressible: stays CTYPE_NONE
				} else {
					fail(LRZGPU_E_INTERNAL);
					return;
				}
			}
			{
				std::lock_guard<std::mutex> lk(mu);
				t_last_enc = now_s();
				enc_busy += t_last_enc - te0;
				enc_wait += te0 - tw0;
			}
			mark_finished(j, true);
		}
	}

	// device -> host through the worker's pinned staging pair (pageable hipMemcpy is ~1 GB/s here)
	static int d2h_staged(void *dst, const void *d_src, size_t bytes, uint8_t *stage[2], hipStream_t s)
	{
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

	void gpu_worker_main()
	{
		if (hipSetDevice(device) != hipSuccess) {
			fail(LRZGPU_E_HIP);
			return;
		}
		hipStream_t s;
		if (make_stream(&s, mask_words, rest_mask) != hipSuccess) {
			fail(LRZGPU_E_HIP);
			return;
		}
		MfWorkspace *ws = nullptr;
		uint8_t *d_stage = nullptr;
		uint8_t *stage[2] = {nullptr, nullptr};
		double per_pos = 16;
		const size_t bufsize = (size_t)sz.stream_bufsize;
		auto cleanup = [&] {
			mf_workspace_destroy(ws);
			if (d_stage)
				(void)hipFree(d_stage);
			for (int k = 0; k < 2; k++)
				if (stage[k])
					(void)hipHostFree(stage[k]);
			(void)hipStreamDestroy(s);
		};
		if (hipHostMalloc((void **)&stage[0], STAGE_BYTES, hipHostMallocDefault) != hipSuccess ||
		    hipHostMalloc((void **)&stage[1], STAGE_BYTES, hipHostMallocDefault) != hipSuccess) {
			fail(LRZGPU_E_NOMEM);
			cleanup();
			return;
		}
		LzmaParams lp;
		const bool lzma_ok = lzma_normalize(lp, sz.level, sz.dict_size, 3, 0, 2, sz.level < 7 ? 32 : 64) == LZ_OK;
		const bool pack = lzma_ok && lp.dict_size <= (1u << 25) && lp.fb <= 127;
		for (;;) {
			Job *j = nullptr;
			{
				std::unique_lock<std::mutex> lk(mu);
				cv_jobs.wait(lk, [&] { return err || (!gpu_queue.empty() && held < held_limit) || (closing && gpu_queue.empty()); });
				if (err || (gpu_queue.empty() && closing)) {
					lk.unlock();
					cleanup();
					return;
				}
				j = gpu_queue.front();
				gpu_queue.pop_front();
				held++; // released in mark_finished
			}
			const double tw0 = now_s();
			const int64_t n = j->ref.len;
			j->done.streamno = j->ref.streamno;
			j->done.s_len = n;
			bool try_backend = !sz.no_compress && n >= 64 && !j->cancelled; // src/stream.c:1633
			if (try_backend && !sz.zstd && !lzma_ok) {
				fail(LRZGPU_E_PARAM);
				cleanup();
				return;
			}
			// block bytes: device view + host copy
			const uint8_t *d_blk = nullptr;
			j->bytes.alloc((size_t)n);
			int rc = 0;
			if (j->ref.streamno == 0) {
				memcpy(j->bytes.data(), j->chunk->stream0.data() + j->ref.off, (size_t)n);
				if (try_backend) {
					if (!d_stage && hipMalloc(&d_stage, bufsize + 256) != hipSuccess)
						rc = LRZGPU_E_NOMEM;
					else if (hipMemcpyAsync(d_stage, j->bytes.data(), (size_t)n, hipMemcpyHostToDevice, s) != hipSuccess ||
						 stream_wait(s) != hipSuccess)
						rc = LRZGPU_E_HIP;
					d_blk = d_stage;
				}
			} else {
				d_blk = j->chunk->d_stream1 + j->ref.off;
				if (n && !j->cancelled && d2h_staged(j->bytes.data(), d_blk, (size_t)n, stage, s) != 0)
					rc = LRZGPU_E_HIP;
			}
			if (rc) {
				fail(rc);
				cleanup();
				return;
			}
			// blocks outside the batched gate (stream 0, > 100 MiB) take the serial gate here
			bool compressible = try_backend;
			if (try_backend && sz.lz4_test && !j->gate_needed) {
				int pct = lrzgpu_lz4_compresses_dev(d_blk, n, sz.threshold, device);
				if (pct < 0) {
					fail(pct);
					cleanup();
					return;
				}
				compressible = pct != 0;
			}
			const double tw1 = now_s();
			double tw2 = tw1;
			if (compressible && !sz.zstd) {
				// match finder on the GPU (runs concurrently with the gate launch of this block)
				unsigned long long total = 0;
				for (int attempt = 0;; attempt++) {
					if (!ws && mf_workspace_create(&ws, bufsize, per_pos) != 0) {
						fail(LRZGPU_E_NOMEM);
						cleanup();
						return;
					}
					int r = mf_run_device(ws, d_blk, (size_t)n, lp.dict_size, (uint32_t)lp.fb, lp.cut(), s, &total, pack, lp.fast);
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
				tw2 = now_s();
				const size_t words = pack ? (size_t)(total / 2) : (size_t)total;
				j->counts.alloc((size_t)n);
				j->pairs.alloc(words);
				j->packed = pack;
				if (d2h_staged(j->counts.data(), ws->counts, (size_t)n, stage, s) != 0 ||
				    (words && d2h_staged(j->pairs.data(), ws->pool_out, words * 4, stage, s) != 0)) {
					fail(LRZGPU_E_HIP);
					cleanup();
					return;
				}
			}
			int act;
			{
				std::lock_guard<std::mutex> lk(mu);
				j->mf_done = true;
				j->compressible_mf = compressible;
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

// Everything the main thread needs to feed blocks to the pipeline while the scan is running.
struct Feeder {
	Pipeline &P;
	hipStream_t ms, ls; // scan/gather stream, current lz4 gate stream
	std::vector<hipStream_t> gate_streams; // gate launches last seconds each: they must overlap one another
	size_t gate_rr = 0;
	std::vector<std::unique_ptr<Job>> owned; // every job ever created (early, final, discarded)
	std::vector<Lz4Batch> batches;
	// gate job descriptors / results live in one arena allocated up front: hipMalloc/hipFree inside the
	// scan would synchronise the whole device (and with it the multi-second gate launches)
	Lz4Job *d_job_arena = nullptr;
	int *d_res_arena = nullptr;
	size_t arena_cap = 0, arena_used = 0;
	int ret = 0;

	Feeder(Pipeline &p) : P(p), ms(nullptr), ls(nullptr) {}

	Job *new_job(ChunkCtx *cc, const BlockRef &br)
	{
		std::unique_ptr<Job> j(new Job());
		j->chunk = cc;
		j->ref = br;
		j->gate_needed = P.sz.lz4_test && !P.sz.no_compress && br.streamno == 1 && br.len >= 64 && br.len <= 100 * 1048576;
		Job *r = j.get();
		owned.push_back(std::move(j));
		return r;
	}

	// queue blocks for the finder and launch their lz4 gate (asynchronously, on `ls`)
	int submit(const std::vector<Job *> &jobs)
	{
		if (jobs.empty())
			return 0;
		{
			std::lock_guard<std::mutex> lk(P.mu);
			for (Job *j : jobs)
				P.gpu_queue.push_back(j);
			P.cv_jobs.notify_all();
		}
		Lz4Batch b;
		std::vector<Lz4Job> lj;
		for (Job *j : jobs)
			if (j->gate_needed) {
				Lz4Job q;
				q.src = j->chunk->d_stream1 + j->ref.off;
				q.src_size = (int)j->ref.len;
				q.dst_capacity = (int)j->ref.len + 1;
				// the container only depends on the verdict (src/stream.c:2325-2380 returns a percentage
				// that is merely printed): let the kernel stop once "compressible" is certain
				q.stop_below = (int)((double)j->ref.len * ((double)P.sz.threshold / 100.0));
				lj.push_back(q);
				b.jobs.push_back(j);
				b.bytes += j->ref.len;
			}
		if (lj.empty())
			return 0;
		if (arena_used + lj.size() > arena_cap)
			return LRZGPU_E_INTERNAL;
		b.d_jobs = d_job_arena + arena_used;
		b.d_res = d_res_arena + arena_used;
		arena_used += lj.size();
		// descriptors go up on the (idle) scan stream: `ls` may still be busy with earlier gate launches
		if (hipMemcpyAsync(b.d_jobs, lj.data(), lj.size() * sizeof(Lz4Job), hipMemcpyHostToDevice, ms) != hipSuccess ||
		    stream_wait(ms) != hipSuccess)
			return LRZGPU_E_HIP;
		ls = gate_streams[gate_rr++ % gate_streams.size()];
		b.timer = new EventTimer(ls);
		int lr = lz4_sizes_device(b.d_jobs, (int)lj.size(), b.d_res, ls);
		b.timer->stop();
		if (lr != 0 || hipEventCreateWithFlags(&b.ev, hipEventBlockingSync | hipEventDisableTiming) != hipSuccess || hipEventRecord(b.ev, ls) != hipSuccess)
			return LRZGPU_E_HIP;
		batches.push_back(std::move(b));
		return 0;
	}

	// collect finished gate launches (all of them when `wait`)
	int poll(bool wait)
	{
		for (size_t k = 0; k < batches.size();) {
			Lz4Batch &b = batches[k];
			hipError_t q = wait ? hipEventSynchronize(b.ev) : hipEventQuery(b.ev);
			if (q == hipErrorNotReady) {
				k++;
				continue;
			}
			if (q != hipSuccess)
				return LRZGPU_E_HIP;
			std::vector<int> res(b.jobs.size());
			if (hipMemcpy(res.data(), b.d_res, res.size() * sizeof(int), hipMemcpyDeviceToHost) != hipSuccess)
				return LRZGPU_E_HIP;
			{
				ProfileStore &ps = ProfileStore::get();
				std::lock_guard<std::mutex> lk(ps.mu);
				ps.p.lz4_ms += b.timer->ms();
				ps.p.lz4_launches++;
				ps.p.lz4_bytes += b.bytes;
			}
			std::vector<Job *> raw;
			{
				std::lock_guard<std::mutex> lk(P.mu);
				for (size_t i = 0; i < b.jobs.size(); i++) {
					b.jobs[i]->lz4_size = res[i];
					b.jobs[i]->lz4_ready = true;
					if (P.route(b.jobs[i]) == 1)
						raw.push_back(b.jobs[i]);
				}
			}
			for (Job *j : raw) {
				if (!j->cancelled)
					P.store_raw(j);
				P.mark_finished(j, true);
			}
			delete b.timer;
			(void)hipEventDestroy(b.ev);
			batches.erase(batches.begin() + (long)k);
		}
		return 0;
	}
};

// memcpy of a result image: a few threads once it is large (one core moves ~5 GB/s)
static void big_copy(uint8_t *dst, const uint8_t *src, size_t n)
{
	const size_t piece = (size_t)64 << 20;
	if (n < 2 * piece) {
		memcpy(dst, src, n);
		return;
	}
	const int nt = n / piece < 8 ? (int)(n / piece) : 8;
	std::vector<std::thread> th;
	for (int t = 0; t < nt; t++) {
		const size_t a = n / nt * t, b = t + 1 == nt ? n : n / nt * (t + 1);
		th.emplace_back([=] { memcpy(dst + a, src + a, b - a); });
	}
	for (auto &x : th)
		x.join();
}

// CPUs this process may burn: the affinity mask, capped by a cgroup CPU quota (v2 cpu.max, v1 cfs_quota_us)
static int usable_cpus()
{
	double n = (double)std::thread::hardware_concurrency();
	cpu_set_t set;
	if (sched_getaffinity(0, sizeof(set), &set) == 0)
		n = (double)CPU_COUNT(&set);
	if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
		char q[64];
		double period = 0;
		if (fscanf(f, "%63s %lf", q, &period) == 2 && strcmp(q, "max") != 0 && period > 0) {
			const double lim = atof(q) / period;
			if (lim > 0 && lim < n)
				n = lim;
		}
		fclose(f);
	} else if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {
		double quota = -1, period = 0;
		if (fscanf(g, "%lf", &quota) != 1)
			quota = -1;
		fclose(g);
		if (FILE *h = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) {
			if (fscanf(h, "%lf", &period) != 1)
				period = 0;
			fclose(h);
		}
		if (quota > 0 && period > 0 && quota / period < n)
			n = quota / period;
	}
	const int r = (int)(n + 0.5);
	return r < 1 ? 1 : r;
}

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
	// host encoders: as asked, else the -p threads capped by the CPUs this process can really use
	// (more runnable threads than the cgroup quota only buys throttling)
	P.n_encoders = ctl->host_threads > 0 ? ctl->host_threads : (ctl->threads > 0 ? ctl->threads : 1);
	if (ctl->host_threads <= 0) {
		const int usable = usable_cpus();
		if (P.n_encoders > usable)
			P.n_encoders = usable;
	}
	if (P.sz.zstd && !ZstdLib::get().ok)
		return LRZGPU_E_PARAM; // --zstd asked for and no libzstd.so.1 on this host
	P.n_gpu_workers = ctl->gpu_slots > 0 ? ctl->gpu_slots : 3;
	P.held_limit = (size_t)P.n_encoders + (size_t)P.n_gpu_workers + 2;
	ctl->stream_bufsize = P.sz.stream_bufsize;
	ctl->dictSize_used = P.sz.dict_size;
	ctl->threads_used = P.sz.threads;
	ctl->st_size = in.n;
	const bool speculate = !getenv("LRZGPU_NO_OVERLAP");
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
				if (hipMemcpyAsync(stage, in.dev + o, k, hipMemcpyDeviceToHost, s) != hipSuccess || stream_wait(s) != hipSuccess) {
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

	P.mask_words = speculate ? scan_cu_words(P.scan_mask, P.rest_mask) : 0;
	const double t0 = now_s();
	double t_scan = 0, t_enq = 0;
	int64_t n_early = 0, n_violations = 0;
	P.start();

	Feeder F(P);
	std::vector<std::unique_ptr<ChunkCtx>> chunks;
	std::vector<Job *> file_order; // every block of the file, in the order the reference writes them
	ScanWorkspace *sw = nullptr;
	int64_t victim_round = 0;
	int64_t len = in.n;
	int ret = 0;
	if (make_stream(&F.ms, P.mask_words, P.scan_mask, true) != hipSuccess)
		ret = LRZGPU_E_HIP;
	for (int k = 0; k < 6 && !ret; k++) {
		hipStream_t gs;
		if (make_stream(&gs, P.mask_words, P.rest_mask) != hipSuccess)
			ret = LRZGPU_E_HIP;
		else
			F.gate_streams.push_back(gs);
	}
	hipStream_t ms = F.ms;
	{
		const int64_t per_chunk = (P.sz.max_chunk < in.n ? P.sz.max_chunk : in.n) / P.sz.stream_bufsize + 8;
		const int64_t nchunks = in.n / (P.sz.max_chunk > 0 ? P.sz.max_chunk : 1) + 2;
		F.arena_cap = (size_t)(per_chunk * nchunks * 2);
		if (!ret && (hipMalloc(&F.d_job_arena, F.arena_cap * sizeof(Lz4Job)) != hipSuccess || hipMalloc(&F.d_res_arena, F.arena_cap * sizeof(int)) != hipSuccess))
			ret = LRZGPU_E_NOMEM;
	}
	uint8_t *d_upload = nullptr; // chunk staging when the input is on the host
	CopyRun *d_runs = nullptr;
	size_t d_runs_cap = 0;
	int pass = 0;
	const int64_t bufsize = P.sz.stream_bufsize;

	// gathers stream-1 bytes [S0, S1) from `runs` (absolute dst offsets)
	auto gather = [&](const uint8_t *d_chunk, ChunkCtx *cc, const std::vector<CopyRun> &runs, int64_t S0, int64_t S1) -> int {
		if (runs.empty() || S1 <= S0)
			return 0;
		if (runs.size() > d_runs_cap) {
			if (d_runs)
				(void)hipFree(d_runs);
			d_runs_cap = runs.size() * 2 + 64;
			if (hipMalloc(&d_runs, d_runs_cap * sizeof(CopyRun)) != hipSuccess)
				return LRZGPU_E_NOMEM;
		}
		if (hipMemcpyAsync(d_runs, runs.data(), runs.size() * sizeof(CopyRun), hipMemcpyHostToDevice, ms) != hipSuccess)
			return LRZGPU_E_HIP;
		EventTimer tg(ms);
		int gr = gather_runs_device(d_chunk, cc->d_stream1, d_runs, (int)runs.size(), S0, S1, ms);
		tg.stop();
		if (gr != 0 || stream_wait(ms) != hipSuccess)
			return LRZGPU_E_HIP;
		ProfileStore &ps = ProfileStore::get();
		std::lock_guard<std::mutex> lk(ps.mu);
		ps.p.gather_ms += tg.ms();
		ps.p.gather_launches++;
		ps.p.gather_bytes += S1 - S0;
		return 0;
	};

	while (!ret && (!pass || len > 0)) { // src/rzip.c:1041
		pass++;
		const int64_t offset = in.n - len;
		const int64_t chunk_size = P.sz.max_chunk < len ? P.sz.max_chunk : len;
		std::unique_ptr<ChunkCtx> cc(new ChunkCtx());
		cc->index = (int)chunks.size();
		cc->size = chunk_size;
		cc->chunk_bytes = chunk_bytes_for(chunk_size);

		const uint8_t *d_chunk = nullptr;
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
		}
		if (hipMalloc(&cc->d_stream1, (size_t)chunk_size + 256) != hipSuccess) {
			ret = LRZGPU_E_NOMEM;
			break;
		}
		if (!sw && scan_workspace_create(&sw, P.sz.rzip_level, P.sz.max_chunk < in.n ? P.sz.max_chunk : in.n) != 0) {
			ret = LRZGPU_E_NOMEM;
			break;
		}

		// ---- speculative early emission while the scan runs -----------------------------------
		int64_t E = 0;          // chunk position up to which stream-1 bytes have been gathered
		int64_t S = 0;          // stream-1 bytes gathered so far
		int64_t seen = 0;       // match records consumed
		int64_t blocks_out = 0; // full stream-1 blocks already submitted
		bool violated = false;
		std::map<int64_t, Job *> early; // stream-1 offset -> job
		std::vector<MatchRec> rec_buf;
		ChunkCtx *ccp = cc.get();

		auto advance = [&](const ScanState &h, int64_t upto, bool final_call, const std::vector<MatchRec> *final_recs) -> int {
			// new records
			const int64_t nrec = final_recs ? (int64_t)final_recs->size() : h.n_records;
			std::vector<CopyRun> runs;
			const int64_t S_before = S;
			if (nrec > seen) {
				const MatchRec *rp;
				if (final_recs)
					rp = final_recs->data() + seen;
				else {
					rec_buf.resize((size_t)(nrec - seen));
					if (hipMemcpy(rec_buf.data(), sw->records + seen, (size_t)(nrec - seen) * sizeof(MatchRec), hipMemcpyDeviceToHost) != hipSuccess)
						return LRZGPU_E_HIP;
					rp = rec_buf.data();
				}
				for (int64_t k = 0; k < nrec - seen && !violated; k++) {
					const MatchRec &r = rp[k];
					if (r.p < E) {
						violated = true; // a match reaches back over bytes already emitted as literals
						break;
					}
					if (E < r.p) {
						runs.push_back(CopyRun{E, S, r.p - E});
						S += r.p - E;
					}
					E = r.p + r.len;
				}
				seen = nrec;
			}
			if (violated)
				return 0;
			int64_t Fp = final_call ? chunk_size : upto - spec_margin();
			if (!final_call && h.cur_len > 0 && h.cur_p < Fp)
				Fp = h.cur_p;
			if (Fp > chunk_size)
				Fp = chunk_size;
			if (Fp > E) {
				if (!runs.empty() && runs.back().src_off + runs.back().len == E)
					runs.back().len += Fp - E;
				else
					runs.push_back(CopyRun{E, S, Fp - E});
				S += Fp - E;
				E = Fp;
			}
			if (S > S_before) {
				int g = gather(d_chunk, ccp, runs, S_before, S);
				if (g)
					return g;
			}
			if (final_call)
				return 0;
			std::vector<Job *> fresh;
			while ((blocks_out + 1) * bufsize <= S) {
				Job *j = F.new_job(ccp, BlockRef{1, blocks_out * bufsize, bufsize});
				early[blocks_out * bufsize] = j;
				fresh.push_back(j);
				blocks_out++;
				n_early++;
			}
			int sr2 = F.submit(fresh);
			if (sr2)
				return sr2;
			return F.poll(false);
		};

		ScanProgressFn progress = nullptr;
		if (speculate)
			progress = [&](const ScanState &h, int64_t upto) -> int { return advance(h, upto, false, nullptr); };

		ScanResult sr;
		int r = scan_chunk_device(sw, d_chunk, chunk_size, P.sz.rzip_level, &victim_round, &sr, ms, progress);
		if (r) {
			ret = r < -50 ? r : (r == -4 ? LRZGPU_E_NOMEM : LRZGPU_E_INTERNAL);
			break;
		}
		t_scan = now_s();
		EmitResult er;
		emit_streams(sr.records, chunk_size, cc->chunk_bytes, sr.crc, &er);
		cc->stream0.swap(er.stream0);
		cc->stream1_len = er.stream1_len;
		if (speculate && !violated) {
			int a = advance(sr.final_state, chunk_size, true, &sr.records);
			if (a) {
				ret = a;
				break;
			}
		}
		if (!speculate || violated || S != er.stream1_len) {
			// (re)build stream 1 from the final run table; early blocks, if any, are void
			if (speculate && violated) {
				ProfileStore &ps = ProfileStore::get();
				std::lock_guard<std::mutex> lk(ps.mu);
				ps.p.spec_rollbacks++;
			}
			if (!early.empty()) {
				n_violations++;
				{
					ProfileStore &ps = ProfileStore::get();
					std::lock_guard<std::mutex> lk(ps.mu);
					ps.p.spec_cancelled_blocks += (int64_t)early.size();
				}
				{
					std::lock_guard<std::mutex> lk(P.mu);
					for (auto &kv : early)
						kv.second->cancelled = true;
				}
				int pr = F.poll(true);
				if (pr) {
					ret = pr;
					break;
				}
				std::unique_lock<std::mutex> lk(P.mu);
				P.cv_done.wait(lk, [&] {
					if (P.err)
						return true;
					for (auto &kv : early)
						if (!kv.second->finished)
							return false;
					return true;
				});
				early.clear();
			}
			int g = gather(d_chunk, ccp, er.runs, 0, er.stream1_len);
			if (g) {
				ret = g;
				break;
			}
		}
		if (hipMemsetAsync(cc->d_stream1 + er.stream1_len, 0, 256, ms) != hipSuccess || stream_wait(ms) != hipSuccess) {
			ret = LRZGPU_E_HIP;
			break;
		}
		// the chunk's blocks in the order the reference flushes them; early blocks are reused
		std::vector<BlockRef> refs;
		block_order(cc->stream0, cc->chunk_bytes, cc->stream1_len, bufsize, &refs);
		std::vector<Job *> fresh;
		for (const BlockRef &br : refs) {
			Job *j = nullptr;
			if (br.streamno == 1 && br.len == bufsize) {
				auto it = early.find(br.off);
				if (it != early.end()) {
					j = it->second;
					early.erase(it);
				}
			}
			if (!j) {
				j = F.new_job(ccp, br);
				fresh.push_back(j);
			}
			file_order.push_back(j);
		}
		if (!early.empty()) { // cannot happen: every early block is a full stream-1 block of the final layout
			ret = LRZGPU_E_INTERNAL;
			break;
		}
		int s2 = F.submit(fresh);
		if (s2) {
			ret = s2;
			break;
		}
		t_enq = now_s();
		chunks.push_back(std::move(cc));
		len -= chunk_size;
	}
	if (ret)
		P.fail(ret);
	else {
		int pr = F.poll(true);
		if (pr) {
			ret = pr;
			P.fail(ret);
		}
	}

	// wait for every block (discarded early ones included: they reference device buffers)
	{
		std::unique_lock<std::mutex> lk(P.mu);
		P.cv_done.wait(lk, [&] {
			if (P.err)
				return true;
			for (auto &j : F.owned)
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
	if (d_runs)
		(void)hipFree(d_runs);
	for (Lz4Batch &b : F.batches) {
		delete b.timer;
		if (b.ev)
			(void)hipEventDestroy(b.ev);
	}
	if (F.d_job_arena)
		(void)hipFree(F.d_job_arena);
	if (F.d_res_arena)
		(void)hipFree(F.d_res_arena);
	if (F.ms)
		(void)hipStreamDestroy(F.ms);
	for (hipStream_t gs : F.gate_streams)
		(void)hipStreamDestroy(gs);
	for (auto &c : chunks)
		if (c->d_stream1) {
			(void)hipFree(c->d_stream1);
			c->d_stream1 = nullptr;
		}
	if (ret)
		return ret;

	// ordered container assembly (one allocation: headers + payloads are known now)
	{
		size_t total = 21 + 16;
		for (auto &c : chunks)
			total += 2 + (size_t)c->chunk_bytes * 7 + 2;
		for (Job *j : file_order)
			total += 1 + 3 * 8 + j->done.payload.size();
		out->reserve(total);
	}
	if (with_magic)
		out->assign(21, 0);
	size_t ji = 0;
	for (size_t ci = 0; ci < chunks.size(); ci++) {
		std::vector<DoneBlock> blocks;
		while (ji < file_order.size() && file_order[ji]->chunk == chunks[ci].get()) {
			blocks.push_back(std::move(file_order[ji]->done));
			ji++;
		}
		write_chunk(out, chunks[ci]->chunk_bytes, ci + 1 == chunks.size(), chunks[ci]->size, blocks);
	}
	out->insert(out->end(), digest, digest + 16);
	memcpy(ctl->hash_resblock, digest, 16);
	if (tracing())
		fprintf(stderr, "lrzgpu driver: scan done %.2f  blocks queued %.2f  last finder %.2f  last encode %.2f  all blocks %.2f  md5 joined %.2f  assembled %.2f s (since start; last chunk); early blocks %lld, redone chunks %lld; worker sums: block copy+gate %.2f finder %.2f lists D2H %.2f, encoders busy %.2f idle %.2f s\n",
			t_scan - t0, t_enq - t0, P.t_last_mf - t0, P.t_last_enc - t0, t_blocks - t0, t_md5 - t0, now_s() - t0,
			(long long)n_early, (long long)n_violations, P.blk_busy, P.mf_busy, P.d2h_busy, P.enc_busy, P.enc_wait);
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
	if (end < 0 && errno == ESPIPE) {
		// a pipe / stdin: the reference spools it into a temporary buffer first (src/lrzip.c:627-922)
		buf->clear();
		std::vector<uint8_t> tmp((size_t)1 << 20);
		for (;;) {
			ssize_t r = read(fd, tmp.data(), tmp.size());
			if (r < 0) {
				if (errno == EINTR)
					continue;
				return LRZGPU_E_IO;
			}
			if (r == 0)
				return 0;
			buf->insert(buf->end(), tmp.data(), tmp.data() + r);
		}
	}
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
	big_copy(*out, o.data(), o.size());
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
	big_copy(*out, o.data(), o.size());
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
	big_copy(*out, o.data(), o.size());
	*out_len = (int64_t)o.size();
	return 0;
}
