// pipeline.h -- the back end of the whole-file driver: blocks (Job) of the chunks being scanned (ChunkCtx) on their way
// through GPU workers (finder, gpu_worker.cpp) and host encoders (parser + range coder, encoder_worker.cpp); queues and
// routing here, the threads' bodies there.  scan_run.cpp feeds it (Feeder) and collects from it.  See driver.cpp for
// the whole picture.
#pragma once
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <pthread.h>
#include <sched.h>
#include <sys/resource.h>

#include <cerrno>
#include <time.h>
#include <sys/mman.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <map>
#include <set>
#include <memory>
#include <string>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/lrzgpu.h"
#include "common.h"
#include "driver.h"
#include "lz4_gate.h"
#include "lzma_enc.h"
#include "lzma_mf.h"
#include "filters.h"
#include "filters_gpu.h"
#include "hashes.h"
#include "md5.h"
#include "pools.h"
#include "profile.h"
#include "rzip_emit.h"
#include "rzip_scan.h"
#include "stream_layer.h"

namespace lrzgpu {

constexpr double kMinPoolPerPos = 4; // list-pool entries per block byte below which no finder workspace is made
void role_cpu_add(int role, double s); // CPU seconds of a pipeline thread, booked to its role (pipeline.cpp)
int control_filter(const lrzgpu_control *c, int *flag, int *delta); // control->filter_flag / delta, checked (driver.cpp)

// literal bytes this far behind the scan count as decided (LRZGPU_SPEC_MARGIN overrides: test hook
// for the roll-back path -- with 0 every match that extends backwards over a segment boundary violates)
inline int64_t spec_margin()
{
	const char *e = getenv("LRZGPU_SPEC_MARGIN"); // read per call: tests flip it inside one process
	return e ? (int64_t)atoll(e) : (int64_t)2 << 20;
}
constexpr size_t STAGE_BYTES = (size_t)32 << 20; // pinned staging piece (uploads, unpinned fall-backs)

inline double now_s()
{
	struct timespec ts;
	clock_gettime(CLOCK_MONOTONIC, &ts);
	return ts.tv_sec + ts.tv_nsec * 1e-9;
}
inline bool tracing()
{
	static int t = getenv("LRZGPU_TRACE") ? 1 : 0;
	return t != 0;
}

// LRZGPU_TRACE=2: one line per block milestone (seconds since the run started) for timeline analysis
extern double g_trace_t0;
extern std::atomic<int> g_trace_events; // read from the environment when a run starts (tests flip it inside one process)
inline bool tracing_events()
{
	return g_trace_events.load(std::memory_order_relaxed) != 0;
}
#define TRACE_EVENT(what, j)                                                                                                        \
	do {                                                                                                                        \
		if (tracing_events())                                                                                               \
			fprintf(stderr, "ev %.3f %s chunk %d stream %d off %lld len %lld\n", now_s() - g_trace_t0, what, (j)->chunk->index, \
				(j)->ref.streamno, (long long)(j)->ref.off, (long long)(j)->ref.len);                                 \
	} while (0)

inline hipError_t make_stream(hipStream_t *s, bool high_priority = false)
{
	const int dev = current_device_or0();
	const int kind = high_priority ? 1 : 0;
	if ((*s = StreamPool::get().take(dev, kind)) != nullptr)
		return hipSuccess;
	hipError_t e = hipErrorUnknown;
	if (high_priority) {
		// a scan stream must never queue behind a multi-second gate/finder kernel: streams share a
		// small pool of hardware queues (GPU_MAX_HW_QUEUES), priority streams get their own
		int lo = 0, hi = 0;
		if (hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess && hi != lo)
			e = hipStreamCreateWithPriority(s, hipStreamNonBlocking, hi);
	}
	if (e != hipSuccess)
		e = hipStreamCreateWithFlags(s, hipStreamNonBlocking);
	if (e == hipSuccess)
		StreamPool::get().created(*s, dev, kind);
	return e;
}

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

struct Job;


struct ChunkCtx {
	int index = 0;
	int64_t offset = 0, size = 0;
	int chunk_bytes = 0;
	bool last = false;
	// input: a view into the caller's device buffer, or an owned copy
	const uint8_t *d_in = nullptr;
	DevBuf in_buf;
	// scan results
	DevBuf stream1; // chunk_size + 256 bytes
	int64_t stream1_len = 0;
	std::vector<uint8_t> stream0;
	int64_t vr_in = 0, vr_out = 0;
	std::vector<std::unique_ptr<Job>> jobs; // every job ever created for this chunk (early, final, discarded)
	std::vector<Job *> file_order;          // the chunk's blocks in the order the reference writes them
	// guarded by Run::mu
	bool input_ready = false, scanned = false;
	bool hash_holds = false;     // the whole-input hash reads the chunk from in_buf: the copy stays until it has
	bool release_wanted = false; // ... and goes then, if the committer has asked for that meanwhile
	double t_scanned = 0;
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
	bool finished = false;
	std::atomic<bool> cancelled{false};
	// ---- early start: the block goes to an encoder before all of it exists.  The finder runs on
	// growing PREFIXES of the block (lists below prefix - fb - 4 are the whole block's: lzma_mf.h block_n), bytes and
	// lists land in the same host arrays stage by stage, the encoder follows through StagedLists::rest.  Guarded by
	// Pipeline::mu.
	bool early = false;
	bool queued = false;         // sits in gpu_queue
	bool in_gpu = false;         // a GPU worker is running a stage of it
	bool held_slot = false;      // counted in Pipeline::held
	bool enc_offered = false;    // sits in enc_queue or is with an encoder: the host side finishes it
	bool with_encoder = false;
	bool retiring = false;       // its encoder is done with it: no further stage
	bool full_requested = false; // the scan has completed the block: the next finder run is the last
	bool full_ready = false;     // whole-block lists and bytes on the host, gate agreed
	bool refused = false;        // the gate said no after an optimistic start: stored
	bool probed = false;         // the first part of the block went through the lz4 gate (a hint: is an early start worth it?)
	bool declined = false;       // ... and looked incompressible: no finder run before the block is complete
	int64_t stage_want = 0;      // bytes of the block gathered so far
	int64_t stage_done = 0;      // prefix the last finished finder run covered
	int64_t valid = 0;           // positions whose lists on the host are final
	int64_t bytes_copied = 0;    // host copy of the block's bytes
	uint64_t words_at_valid = 0; // words of pairs[] in front of position `valid`
	std::vector<std::unique_ptr<RawBuf<uint32_t>>> old_pairs; // outgrown list arrays an encoder may still read
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
	lrzgpu_control *ctl = nullptr;
	Sizing sz;
	int device = 0;
	int filter_flag = 0, filter_delta = 0; // control->filter_flag / delta: every literal block through this filter first
	int n_gpu_workers = 2, n_encoders = 1;
	double mf_per_pos = 16; // list-pool entries per block byte the finder workspaces start with (less for blocks that only fit so)
	std::atomic<int> err{0};          // first failure; read by every thread of the run
	std::function<void()> on_fail;    // wakes the run's own waiters (reader, scanners, committer)

	std::mutex mu;
	std::condition_variable cv_jobs, cv_enc, cv_done;
	std::condition_variable cv_rest; // early jobs: a stage arrived / the gate spoke / a worker left the job / cancelled
	std::deque<Job *> gpu_queue; // blocks waiting for a GPU worker
	std::deque<Job *> enc_queue; // blocks with match lists and a positive gate, waiting for a host encoder
	size_t held = 0;             // blocks holding host match lists (bounds host memory)
	size_t held_limit = 4;
	bool closing = false;
	double t_last_mf = 0, t_last_enc = 0;
	double mf_busy = 0, d2h_busy = 0, blk_busy = 0, enc_busy = 0, enc_wait = 0;
	std::vector<std::thread> threads;
	// early start
	int early_mode = 1;         // 0 off, 1 while encoders have nothing to do, 2 every block (LRZGPU_EARLY_START; tests force 2)
	int64_t early_first = 0;    // bytes of a block that must be there before its first finder run
	int64_t early_step = 0;     // ... and between two runs
	bool early_split = true;    // a complete block met by idle encoders gets a short first finder run too
	int enc_waiting = 0;        // encoder threads with nothing to do
	int early_unclaimed = 0;    // early jobs no encoder has taken yet
	double rest_wait = 0, t_first_enc = 0;
	int64_t n_early_jobs = 0, n_early_stages = 0;

	// the waiting block that comes first in the FILE (chunks are scanned side by side and their blocks arrive
	// interleaved): chunks then complete one after the other and are laid out / written while later ones are
	// still being encoded, instead of all at the very end
	static bool file_order_before(const Job *a, const Job *b)
	{
		return a->chunk->index < b->chunk->index || (a->chunk->index == b->chunk->index && a->ref.streamno == b->ref.streamno && a->ref.off < b->ref.off);
	}
	// next block for an encoder (mu held): withdrawn ones first (dropping them is what their chunk's scanner waits for),
	// then complete blocks in file order, a block that is still arriving only when nothing else waits
	Job *take_enc()
	{
		size_t best = 0;
		auto rank = [](const Job *j) { return j->cancelled ? 0 : ((j->early && !j->full_ready && !j->refused) ? 2 : 1); };
		for (size_t i = 1; i < enc_queue.size(); i++) {
			const Job *a = enc_queue[i], *b = enc_queue[best];
			const int ra = rank(a), rb = rank(b);
			if (ra < rb || (ra == rb && file_order_before(a, b)))
				best = i;
		}
		Job *j = enc_queue[best];
		enc_queue.erase(enc_queue.begin() + (long)best);
		return j;
	}
	// next block for a GPU worker (mu held), nullptr if none may be taken now: stages of early blocks first (an encoder
	// is following them), then file order; a block that holds no host buffers yet only below the limit
	Job *take_gpu()
	{
		size_t best = gpu_queue.size();
		for (size_t i = 0; i < gpu_queue.size(); i++) {
			const Job *a = gpu_queue[i];
			if (!a->held_slot && held >= held_limit)
				continue;
			if (best == gpu_queue.size()) {
				best = i;
				continue;
			}
			const Job *b = gpu_queue[best];
			if (a->early != b->early ? a->early : file_order_before(a, b))
				best = i;
		}
		if (best == gpu_queue.size())
			return nullptr;
		Job *j = gpu_queue[best];
		gpu_queue.erase(gpu_queue.begin() + (long)best);
		return j;
	}
	void enqueue_gpu(Job *j) // mu held
	{
		if (j->early) {
			j->full_requested = true; // (the only way an early job comes here again: its block is complete)
			j->stage_want = j->ref.len;
			if (j->queued || j->in_gpu || j->retiring || j->finished)
				return;
		}
		j->queued = true;
		gpu_queue.push_back(j);
	}
	// the scanner has gathered `have` bytes of an early block (mu not held)
	void stage(Job *j, int64_t have)
	{
		std::lock_guard<std::mutex> lk(mu);
		if (j->full_requested || j->finished || j->retiring || j->cancelled)
			return;
		j->stage_want = have;
		if (j->queued || j->in_gpu)
			return; // the worker looks again when it is through
		if (have - j->stage_done >= (j->stage_done ? early_step : early_first)) {
			j->queued = true;
			gpu_queue.push_back(j);
			cv_jobs.notify_all();
		}
	}
	// should a block be started early now? (mu not held)
	bool want_early()
	{
		if (early_mode == 2)
			return true;
		if (early_mode == 0)
			return false;
		std::lock_guard<std::mutex> lk(mu);
		return enc_waiting > early_unclaimed && enc_queue.empty();
	}

	void fail(int e)
	{
		{
			std::lock_guard<std::mutex> lk(mu);
			int none = 0;
			err.compare_exchange_strong(none, e);
			cv_jobs.notify_all();
			cv_enc.notify_all();
			cv_done.notify_all();
			cv_rest.notify_all();
		}
		if (on_fail)
			on_fail();
	}
	int error() const { return err.load(); }

	void finish_locked(Job *j) // mu held; the job's buffers have been given back
	{
		if (j->held_slot) {
			j->held_slot = false;
			held--;
			cv_jobs.notify_all();
		}
		if (j->early && !j->with_encoder)
			early_unclaimed--;
		j->finished = true;
		cv_done.notify_all();
		cv_rest.notify_all();
	}
	void mark_finished(Job *j, bool)
	{
		j->bytes.release();
		j->counts.release();
		j->pairs.release();
		j->old_pairs.clear();
		std::lock_guard<std::mutex> lk(mu);
		finish_locked(j);
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
		if (j->enc_offered) {
			// started early: an encoder has the block (or will take it from the queue) and finishes it either way
			if (compressible)
				j->full_ready = true;
			else {
				j->refused = true;
				TRACE_EVENT("refused_late", j);
			}
			cv_rest.notify_all();
			return 0;
		}
		if (compressible) {
			if (j->early) {
				j->full_ready = true;
				j->enc_offered = true;
			}
			enc_queue.push_back(j);
			cv_enc.notify_one();
			return 0;
		}
		return 1;
	}

	// ---- the encoder's side of an early block ------------------------------------------------------------
	struct RestCtx {
		Pipeline *P;
		Job *j;
		int64_t seen; // the limit the parser was told last
		MatchLists ml;
		double waited = 0;
	};
	// encoder_worker.cpp: StagedLists::rest of an early block, the end of an encoder's part in one, the encoder thread
	static const MatchLists *rest_cb(void *ctx, size_t *valid);
	void retire(Job *j);
	void encoder_main();
	// gpu_worker.cpp: the GPU worker thread (finder runs, list copies, the stages of early blocks)
	struct GpuWorker;
	static int d2h(void *dst, bool dst_pinned, const void *d_src, size_t bytes, uint8_t *stage[2], hipStream_t s);
	int gpu_open(GpuWorker &w);
	void gpu_close(GpuWorker &w);
	int gpu_run_finder(GpuWorker &w, const uint8_t *d_blk, size_t n, size_t block_n, unsigned long long *total);
	int gpu_early_probe(GpuWorker &w, Job *j, const uint8_t *d_blk, int64_t P);
	int gpu_early_stage(GpuWorker &w, Job *j);
	int gpu_whole_block(GpuWorker &w, Job *j, double tw0);
	void gpu_worker_main();

	// a thread body: nothing may escape it (std::terminate), failures become the pipeline's error; the CPU time the
	// thread burnt is booked to its role (0 encoders, 1 GPU workers, 2 scanners, 3 hash, 4 reader)
	template <typename F> void guarded(F &&f, int role = -1)
	{
		try {
			f();
		} catch (const std::bad_alloc &) {
			fail(LRZGPU_E_NOMEM);
		} catch (...) {
			fail(LRZGPU_E_INTERNAL);
		}
		struct timespec ts;
		if (role >= 0 && clock_gettime(CLOCK_THREAD_CPUTIME_ID, &ts) == 0)
			role_cpu_add(role, ts.tv_sec + ts.tv_nsec * 1e-9);
	}

	void start()
	{
		for (int i = 0; i < n_gpu_workers; i++)
			threads.emplace_back([this] { guarded([this] { gpu_worker_main(); }, 1); });
		for (int i = 0; i < n_encoders; i++)
			threads.emplace_back([this] { guarded([this] { encoder_main(); }, 0); });
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

	// mark jobs void and wait until no thread touches them (or their chunk's device buffers) any more
	void cancel_and_wait(const std::vector<Job *> &jobs)
	{
		std::unique_lock<std::mutex> lk(mu);
		for (Job *j : jobs) {
			j->cancelled = true;
			// an early block no encoder has taken yet leaves the queues here (the encoders may all be busy for seconds);
			// one that is in a finder run is ended by its worker, one that is with an encoder by the encoder.  A block
			// that was never offered to the encoders (its first part looked incompressible to the gate, or its first
			// finder run has not happened yet) and sits in no queue between two stages has nobody else to end it.
			if (j->early && !j->with_encoder && !j->finished) {
				if (j->enc_offered)
					for (size_t i = 0; i < enc_queue.size(); i++)
						if (enc_queue[i] == j) {
							enc_queue.erase(enc_queue.begin() + (long)i);
							j->enc_offered = false;
							break;
						}
				if (!j->enc_offered) {
					if (j->queued) {
						for (size_t i = 0; i < gpu_queue.size(); i++)
							if (gpu_queue[i] == j) {
								gpu_queue.erase(gpu_queue.begin() + (long)i);
								break;
							}
						j->queued = false;
					}
					if (!j->in_gpu) {
						j->bytes.release();
						j->counts.release();
						j->pairs.release();
						j->old_pairs.clear();
						finish_locked(j);
					}
				}
			}
		}
		cv_rest.notify_all();
		cv_done.wait(lk, [&] {
			if (err)
				return true;
			for (Job *j : jobs)
				if (!j->finished)
					return false;
			return true;
		});
	}
};

// What a scanner thread needs to feed blocks to the pipeline while its scan is running.
struct Feeder {
	Pipeline &P;
	hipStream_t ms = nullptr;                // scan/gather stream
	std::vector<hipStream_t> gate_streams;   // gate launches last seconds each: they must overlap one another
	size_t gate_rr = 0;
	std::vector<Lz4Batch> batches;
	// gate job descriptors / results live in arenas allocated outside the scan: hipMalloc/hipFree inside
	// it would synchronise the whole device (and with it the multi-second gate launches)
	DevBuf arena;
	size_t arena_cap = 0, arena_used = 0;

	explicit Feeder(Pipeline &p) : P(p) {}

	int reserve(size_t descriptors)
	{
		if (arena_cap - arena_used >= descriptors)
			return 0;
		if (!batches.empty()) // descriptors of launches in flight live in the current arena
			return LRZGPU_E_INTERNAL;
		arena.release();
		arena_cap = descriptors < 4096 ? 4096 : descriptors;
		arena_used = 0;
		if (!arena.alloc(arena_cap * (sizeof(Lz4Job) + sizeof(int)) + 64, P.device))
			return LRZGPU_E_NOMEM;
		return 0;
	}

	Job *new_job(ChunkCtx *cc, const BlockRef &br)
	{
		std::unique_ptr<Job> j(new Job());
		j->chunk = cc;
		j->ref = br;
		j->gate_needed = P.sz.lz4_test && !P.sz.no_compress && br.streamno == 1 && br.len >= 64 && br.len <= 100 * 1048576;
		Job *r = j.get();
		cc->jobs.push_back(std::move(j));
		return r;
	}

	// queue blocks for the finder and launch their lz4 gate (asynchronously)
	int submit(const std::vector<Job *> &jobs)
	{
		if (jobs.empty())
			return 0;
		{
			std::lock_guard<std::mutex> lk(P.mu);
			for (Job *j : jobs) {
				P.enqueue_gpu(j);
				TRACE_EVENT("submit", j);
			}
			P.cv_jobs.notify_all();
		}
		Lz4Batch b;
		std::vector<Lz4Job> lj;
		for (Job *j : jobs)
			if (j->gate_needed) {
				Lz4Job q;
				q.src = j->chunk->stream1.p + j->ref.off;
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
		b.d_jobs = (Lz4Job *)arena.p + arena_used;
		b.d_res = (int *)(arena.p + arena_cap * sizeof(Lz4Job)) + arena_used;
		arena_used += lj.size();
		// descriptors go up on the (idle) scan stream: a gate stream may still be busy with earlier launches
		if (hipMemcpyAsync(b.d_jobs, lj.data(), lj.size() * sizeof(Lz4Job), hipMemcpyHostToDevice, ms) != hipSuccess ||
		    stream_wait(ms) != hipSuccess)
			return LRZGPU_E_HIP;
		hipStream_t ls = gate_streams[gate_rr++ % gate_streams.size()];
		b.timer = new EventTimer(ls);
		int lr = lz4_sizes_device(b.d_jobs, (int)lj.size(), b.d_res, ls);
		b.timer->stop();
		if (lr != 0 || hipEventCreateWithFlags(&b.ev, hipEventDisableTiming) != hipSuccess || hipEventRecord(b.ev, ls) != hipSuccess) {
			delete b.timer;
			return LRZGPU_E_HIP;
		}
		batches.push_back(std::move(b));
		return 0;
	}

	// collect finished gate launches (all of them when `wait`)
	int poll(bool wait)
	{
		for (size_t k = 0; k < batches.size();) {
			Lz4Batch &b = batches[k];
			hipError_t q = wait ? event_wait(b.ev) : hipEventQuery(b.ev);
			if (q == hipErrorNotReady) {
				k++;
				continue;
			}
			if (q != hipSuccess)
				return LRZGPU_E_HIP;
			std::vector<int> res(b.jobs.size());
			if (d2h_pageable(res.data(), b.d_res, res.size() * sizeof(int), ms) != hipSuccess)
				return LRZGPU_E_HIP;
			{
				ProfileStore &ps = ProfileStore::get();
				std::lock_guard<std::mutex> lk(ps.mu);
				ps.p.lz4_ms += b.timer->ms_noted(ps, PK_LZ4);
				ps.p.lz4_launches++;
				ps.p.lz4_bytes += b.bytes;
			}
			std::vector<Job *> raw;
			{
				std::lock_guard<std::mutex> lk(P.mu);
				for (size_t i = 0; i < b.jobs.size(); i++) {
					b.jobs[i]->lz4_size = res[i];
					b.jobs[i]->lz4_ready = true;
					TRACE_EVENT("gate_done", b.jobs[i]);
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

	void destroy()
	{
		for (Lz4Batch &b : batches) {
			if (b.ev) {
				(void)event_wait(b.ev);
				(void)hipEventDestroy(b.ev);
			}
			delete b.timer;
		}
		batches.clear();
		if (ms)
			StreamPool::get().give(ms);
		for (hipStream_t gs : gate_streams)
			StreamPool::get().give(gs);
		ms = nullptr;
		gate_streams.clear();
		arena.release();
	}
};

// CPUs this process may burn: the affinity mask, capped by a cgroup CPU quota (v2 cpu.max, v1 cfs_quota_us)
inline int usable_cpus()
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

inline int write_all(int fd, const uint8_t *p, size_t n)
{
	while (n) {
		ssize_t w = write(fd, p, n > ((size_t)1 << 30) ? ((size_t)1 << 30) : n);
		if (w < 0 && errno == EINTR)
			continue;
		if (w <= 0)
			return LRZGPU_E_IO;
		p += w;
		n -= (size_t)w;
	}
	return 0;
}

inline int pread_all(int fd, uint8_t *p, size_t n, int64_t off)
{
	while (n) {
		ssize_t r = pread(fd, p, n > ((size_t)1 << 30) ? ((size_t)1 << 30) : n, (off_t)off);
		if (r < 0 && errno == EINTR)
			continue;
		if (r <= 0)
			return LRZGPU_E_IO;
		p += r;
		n -= (size_t)r;
		off += r;
	}
	return 0;
}


} // namespace lrzgpu
