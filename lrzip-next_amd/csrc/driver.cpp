// driver.cpp -- whole-file compress driver: the rzip_fd() control flow (reference src/rzip.c:922-1264)
// over the GPU stages, with chunks scanned concurrently and the per-block back end pipelined between
// the GPU and host threads.
//
//   reader           one thread, chunk after chunk in file order: makes the chunk's bytes resident in HBM
//                    (a view of a device buffer, or pread()/memcpy into pinned pieces + H2D) -- the
//                    reference maps one chunk at a time (src/rzip.c:1057-1107)
//   scanners         `scan_slots` threads, each with a scan stream + resolver workspace + gate streams.
//                    rzip chunks are independent units of the format (own table, own CRC; src/rzip.c:
//                    599-626); the resolver of one chunk is ONE wavefront, so several chunks are
//                    resolved side by side on different CUs.  The only state that crosses a chunk boundary
//                    is insert_hash()'s static victim_round (src/rzip.c:308): a chunk that starts before
//                    its predecessor has finished assumes the value the predecessor is expected to leave
//                    and is scanned again if that turns out wrong.
//                    While a chunk is scanned the literal bytes that are already decided are gathered
//                    (K4) and every stream-1 block they complete is handed to the back end AT ONCE -- the
//                    reference overlaps the same way through flush_buffer() (src/rzip.c:229-246).
//                    "Decided" = behind every emitted match, or more than SPEC_MARGIN behind the scan
//                    position; a later match reaching back over such bytes is detected and the chunk's
//                    early blocks are then thrown away and redone (tests force it with
//                    LRZGPU_SPEC_MARGIN / LRZGPU_SEG_BYTES).
//   lz4 gate         one wavefront per block, launched per group of new blocks on its own stream
//   GPU workers      `gpu_slots` threads, each with a HIP stream + match-finder workspace: finder for
//                    block k+1 while block k is parsed on the host; lists land in pinned host buffers
//   host encoders    `host_threads` threads (default: the CPUs the process may use): LZMA parser +
//                    range coder (lzma_enc.cpp); all stream waits are blocking-sync events
//   committer        the calling thread: validates the victim_round chain in chunk order, waits for the
//                    chunk's blocks, lays the chunk out (stream_layer.cpp; the file order of the blocks is
//                    block_order()'s replay of the reference flushes, whatever order they finished in)
//                    and hands it to the sink (memory image or fd) -- one chunk of output resident at a time
//   md5              whole-input MD5 on its own thread (serial by construction)
//
// Output bytes depend only on (input, control parameters), never on thread counts or timing here.
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

using namespace lrzgpu;

// CPU seconds burnt by the threads of the whole-file pipeline, by role, since the last lrzgpu_profile_reset()
static std::mutex g_role_mu;
static double g_role_cpu[8] = {0, 0, 0, 0, 0, 0, 0, 0};
static void role_cpu_add(int role, double s)
{
	std::lock_guard<std::mutex> lk(g_role_mu);
	g_role_cpu[role & 7] += s;
}

namespace lrzgpu {
int control_filter(const lrzgpu_control *c, int *flag, int *delta);
} // namespace lrzgpu
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
	c->hash_code = 1; // MD5 (src/lrzip.c:1842); -H <n> of the reference's command line = this field, filter options =
	                  // filter_flag / delta (0: none): per run, where the reference keeps them (rzip_control)
	c->fd_out = -1;
}

// the filter options of the reference's command line (--x86 ... --delta=N, src/main.c:612-660) as they stand in a control
static int check_filter(int filter_flag, int delta)
{
	if (filter_flag != 0 && !lrzgpu::filter_supported(filter_flag, delta))
		return LRZGPU_E_PARAM;
	if (filter_flag == lrzgpu::FILTER_DELTA && delta > 16 && delta % 16)
		return LRZGPU_E_PARAM; // magic[16] codes 1..16, 32, 48 ... 256 only (src/lrzip.c:148-156)
	return 0;
}
namespace lrzgpu {
int control_filter(const lrzgpu_control *c, int *flag, int *delta)
{
	*flag = c->filter_flag;
	*delta = c->filter_flag == FILTER_DELTA ? c->delta : 0;
	return check_filter(*flag, *delta);
}
} // namespace lrzgpu

// (lrzgpu_hash.h) CPU seconds of the pipeline's threads by role: 0 encoders (parser + range coder / zstd), 1 GPU workers
// (block copies, finder launches, list copies), 2 scanners, 3 the whole-input hash, 4 the reader; reset != 0 clears
extern "C" void lrzgpu_profile_cpu(double out[8], int reset)
{
	std::lock_guard<std::mutex> lk(g_role_mu);
	for (int k = 0; k < 8; k++) {
		if (out)
			out[k] = g_role_cpu[k];
		if (reset)
			g_role_cpu[k] = 0;
	}
}

constexpr double kMinPoolPerPos = 4; // list-pool entries per block byte below which no finder workspace is made

extern "C" void lrzgpu_trim(void)
{
	WorkspacePool::get().trim();
	DevicePool::get().trim();
	HostPool::get().trim();
}

// The largest LZMA block (control->stream_bufsize, what lrzgpu_plan() reports) a run on `device` can take: the match
// finder needs ~110 B of sort keys, links and tree nodes plus its list pools per block byte, all of it resident for the
// walk (the reference's finder: ~11.5 B per dictionary byte on the host, src/util.c:108-131), so the device's memory
// bounds the block -- beside one chunk's input, its literal stream and a scan workspace.  <= 0: no such device.
extern "C" int64_t lrzgpu_max_block_bytes(int device)
{
	int cur = 0;
	if (hipGetDevice(&cur) != hipSuccess || hipSetDevice(device) != hipSuccess)
		return 0;
	size_t free_b = 0, total = 0;
	const hipError_t e = hipMemGetInfo(&free_b, &total);
	(void)hipSetDevice(cur);
	if (e != hipSuccess)
		return 0;
	// workspace(n) + 2 n (the block's chunk is at least the block: input + literal stream) + scan workspace + margin <= total
	const double per_byte = 110.0 + 8.0 * kMinPoolPerPos + 2.0;
	const double room = (double)total - (double)DeviceBudget::margin() - (double)((size_t)4 << 30) - (double)((size_t)64 << 20);
	const double n = room / per_byte;
	const double cap = 4294967295.0 - 65536.0; // (and the 32-bit positions of the format's encoder)
	return n <= 0 ? 0 : (int64_t)(n < cap ? n : cap);
}

// lrzgpu_trim() plus the parked streams: for a caller that is about to exit (profilers want every queue closed).
// Not for use between files: a later allocation in the same process can trip the runtime over the closed queues
// (pools.h, DeviceBudget).
extern "C" void lrzgpu_shutdown(void)
{
	lrzgpu_trim();
	StreamPool::get().destroy_idle();
}

namespace lrzgpu {

// literal bytes this far behind the scan count as decided (LRZGPU_SPEC_MARGIN overrides: test hook
// for the roll-back path -- with 0 every match that extends backwards over a segment boundary violates)
static int64_t spec_margin()
{
	const char *e = getenv("LRZGPU_SPEC_MARGIN"); // read per call: tests flip it inside one process
	return e ? (int64_t)atoll(e) : (int64_t)2 << 20;
}
constexpr size_t STAGE_BYTES = (size_t)32 << 20; // pinned staging piece (uploads, unpinned fall-backs)

static double now_s()
{
	struct timespec ts;
	clock_gettime(CLOCK_MONOTONIC, &ts);
	return ts.tv_sec + ts.tv_nsec * 1e-9;
}
static bool tracing()
{
	static int t = getenv("LRZGPU_TRACE") ? 1 : 0;
	return t != 0;
}

// LRZGPU_TRACE=2: one line per block milestone (seconds since the run started) for timeline analysis
static double g_trace_t0 = 0;
static std::atomic<int> g_trace_events{0}; // read from the environment when a run starts (tests flip it inside one process)
static bool tracing_events()
{
	return g_trace_events.load(std::memory_order_relaxed) != 0;
}
#define TRACE_EVENT(what, j)                                                                                                        \
	do {                                                                                                                        \
		if (tracing_events())                                                                                               \
			fprintf(stderr, "ev %.3f %s chunk %d stream %d off %lld len %lld\n", now_s() - g_trace_t0, what, (j)->chunk->index, \
				(j)->ref.streamno, (long long)(j)->ref.off, (long long)(j)->ref.len);                                 \
	} while (0)

static hipError_t make_stream(hipStream_t *s, bool high_priority = false)
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
	// StagedLists::rest: blocks until the finder has covered more of the block (or all of it and the gate agreed)
	static const MatchLists *rest_cb(void *ctx, size_t *valid)
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
	void retire(Job *j)
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
	void encoder_main()
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

	// device -> host.  Pinned destinations take the DMA directly; pageable ones go through the worker's
	// pinned staging pair (a pageable hipMemcpy is ~1 GB/s here)
	static int d2h(void *dst, bool dst_pinned, const void *d_src, size_t bytes, uint8_t *stage[2], hipStream_t s)
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

	void gpu_worker_main()
	{
		if (hipSetDevice(device) != hipSuccess) {
			fail(LRZGPU_E_HIP);
			return;
		}
		hipStream_t s;
		if (make_stream(&s) != hipSuccess) {
			fail(LRZGPU_E_HIP);
			return;
		}
		MfWorkspace *ws = nullptr;
		double ws_per_pos = 0;
		DevBuf d_stage, d_scratch, d_probe;
		uint8_t *stage[2] = {nullptr, nullptr};
		double per_pos = mf_per_pos;
		const size_t bufsize = (size_t)sz.stream_bufsize;
		const bool want_pinned = true; // lists and block bytes land in pinned host buffers from the pool
		auto cleanup = [&] {
			WorkspacePool::get().give_mf(ws, ws_per_pos, device);
			ws = nullptr;
			d_stage.release();
			d_scratch.release();
			d_probe.release();
			for (int k = 0; k < 2; k++)
				if (stage[k])
					(void)hipHostFree(stage[k]);
			StreamPool::get().give(s);
		};
		if (hipHostMalloc((void **)&stage[0], STAGE_BYTES, hipHostMallocDefault) != hipSuccess ||
		    hipHostMalloc((void **)&stage[1], STAGE_BYTES, hipHostMallocDefault) != hipSuccess) {
			fail(LRZGPU_E_NOMEM);
			cleanup();
			return;
		}
		LzmaParams lp;
		const bool lzma_ok = lzma_normalize(lp, sz.level, sz.dict_size, 3, 0, 2, sz.level < 7 ? 32 : 64) == LZ_OK;
		// lists with the tail flag; one word per pair when the format allows it (lzma_mf.hip k_gather)
		const bool pack = lzma_ok && lp.dict_size <= (1u << 25) && lp.fb <= 65;
		// the finder on d_blk[0..n), a prefix of a block of block_n bytes (0: the block itself); grows the pool when the
		// data needs more list entries than it holds
		auto run_finder = [&](const uint8_t *d_blk, size_t n, size_t block_n, unsigned long long *total) -> int {
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
		};
		// ---- one finder run of an early block: the prefix that is there, or the whole block ----
		auto early_stage = [&](Job *j) -> int {
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
			// The gate's verdict is taken for granted when a block is started early; on data it refuses (random bytes:
			// BASELINE configs[4]) that would be a finder run and an encoder per block for nothing.  So the first part of
			// the block goes through the gate once, as a hint: if lz4 finds nothing in it, the block waits for its
			// completion like any other (the verdict that counts is the one on the whole block, as ever).
			if (!full && !j->cancelled && sz.lz4_test && !j->probed) {
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
					int fr = run_finder(d_blk, (size_t)P, full ? 0 : (size_t)n, &total);
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
		};
		for (;;) {
			Job *j = nullptr;
			{
				std::unique_lock<std::mutex> lk(mu);
				cv_jobs.wait(lk, [&] { return err || (closing && gpu_queue.empty()) || (j = take_gpu()) != nullptr; });
				if (!j) {
					lk.unlock();
					cleanup();
					return;
				}
				j->queued = false;
				j->in_gpu = true;
				if (!j->held_slot) {
					j->held_slot = true;
					held++; // released in finish_locked
				}
			}
			const double tw0 = now_s();
			TRACE_EVENT("gpu_start", j);
			const int64_t n = j->ref.len;
			j->done.streamno = j->ref.streamno;
			j->done.s_len = n;
			bool try_backend = !sz.no_compress && n >= 64 && !j->cancelled; // src/stream.c:1633
			if (try_backend && !sz.zstd && !lzma_ok) {
				fail(LRZGPU_E_PARAM);
				cleanup();
				return;
			}
			if (j->early) {
				// whichever way a run ends, the job must not stay marked "in a finder run": its encoder waits for that
				// mark to clear before it lets go of the block's buffers (retire), and would wait for ever
				auto left_the_gpu = [&] {
					std::lock_guard<std::mutex> lk(mu);
					j->in_gpu = false;
					cv_rest.notify_all();
				};
				int er;
				try {
					er = early_stage(j);
				} catch (...) {
					left_the_gpu();
					throw;
				}
				TRACE_EVENT("gpu_end", j);
				if (er) {
					left_the_gpu();
					fail(er);
					cleanup();
					return;
				}
				continue;
			}
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
				int fr = run_finder(d_blk, (size_t)n, 0, &total);
				if (fr) {
					fail(fr);
					cleanup();
					return;
				}
				tw2 = now_s();
				const size_t words = pack ? (size_t)(total / 2) : (size_t)total;
				j->counts.alloc((size_t)n, want_pinned);
				j->pairs.alloc(words, want_pinned);
				j->packed = pack;
				if (d2h(j->counts.data(), j->counts.pinned, ws->counts, (size_t)n, stage, s) != 0 ||
				    (words && d2h(j->pairs.data(), j->pairs.pinned, ws->pool_out, words * 4, stage, s) != 0)) {
					fail(LRZGPU_E_HIP);
					cleanup();
					return;
				}
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
		}
	}

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

static int write_all(int fd, const uint8_t *p, size_t n)
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

static int pread_all(int fd, uint8_t *p, size_t n, int64_t off)
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

// ---- one compress run ------------------------------------------------------------------------------
struct Run {
	lrzgpu_control *ctl;
	const CompressSource &in;
	CompressSink &out;
	const ChunkSelect *sel;
	Pipeline P;
	std::vector<std::unique_ptr<ChunkCtx>> chunks; // outlive every thread of the run
	std::vector<int> mine;                         // indices into `chunks` this run compresses, ascending
	std::mutex mu;
	std::condition_variable cv;
	size_t next_scan = 0;   // position in `mine` the next free scanner takes
	size_t committed = 0;   // chunks of `mine` already laid out: the reader stays a bounded distance ahead
	int scan_slots = 1;
	bool speculate = true;
	int64_t n_early = 0, n_violations = 0, n_rescans = 0;
	double t0 = 0;

	Run(lrzgpu_control *c, const CompressSource &i, CompressSink &o, const ChunkSelect *s) : ctl(c), in(i), out(o), sel(s) {}

	void fail(int e) { P.fail(e); } // (P.on_fail wakes this run's waiters)

	// ---- readers: chunk bytes into HBM ---------------------------------------------------------------------
	// With the input in HBM already one reader hands out views (or device-to-device copies).  A file or a host buffer
	// is read by as many readers as there are scanners, ALL of them on the same chunk: reader t of n takes the pieces
	// t, t + n, ... of every chunk, in file order.  One thread moves ~6 GB/s out of the page cache through its two
	// pinned pieces, and a scanner can only start on a chunk that is there completely (a match may run to the chunk's
	// end): read by one thread, the eighth chunk of the headline file was ready 2.7 s after the first -- and its scan
	// that much later; read by eight, chunk k is ready 45 ms after chunk k - 1.
	int n_readers = 1;
	struct ReadState { // per chunk of `mine`, guarded by mu
		int arrived = 0; // readers that have their pieces of the chunk in HBM
		int rc = 0;
		bool allocated = false;
	};
	std::vector<ReadState> read_state;
	void reader_main(int t)
	{
		if (hipSetDevice(P.device) != hipSuccess) {
			fail(LRZGPU_E_HIP);
			return;
		}
		hipStream_t s = nullptr;
		RawBuf<uint8_t> stage_buf[2]; // pinned, from the pool (a run after the first finds them there)
		uint8_t *stage[2] = {nullptr, nullptr};
		hipEvent_t done[2] = {nullptr, nullptr};
		auto cleanup = [&] {
			for (int q = 0; q < 2; q++)
				if (done[q])
					(void)hipEventDestroy(done[q]);
			if (s)
				StreamPool::get().give(s);
		};
		if (make_stream(&s) != hipSuccess) {
			fail(LRZGPU_E_HIP);
			return;
		}
		const bool pieces = !in.dev && !in.dev_chunks; // host memory or a file: through pinned pieces
		if (pieces) {
			stage_buf[0].alloc(STAGE_BYTES, true);
			stage_buf[1].alloc(STAGE_BYTES, true);
			stage[0] = stage_buf[0].data();
			stage[1] = stage_buf[1].data();
			if (hipEventCreateWithFlags(&done[0], hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&done[1], hipEventDisableTiming) != hipSuccess) {
				fail(LRZGPU_E_HIP);
				cleanup();
				return;
			}
		}
		bool used[2] = {false, false};
		int k = 0;
		for (size_t m = 0; m < mine.size(); m++) {
			ChunkCtx *cc = chunks[(size_t)mine[m]].get();
			int rc = 0;
			{
				// at most scan_slots + 1 chunks ahead of the committer hold input copies; the first reader to arrive
				// sets the chunk's buffer up for all of them
				std::unique_lock<std::mutex> lk(mu);
				cv.wait(lk, [&] { return P.err || m < committed + (size_t)scan_slots + 1; });
				if (P.err)
					break;
				ReadState &rs = read_state[m];
				if (!rs.allocated) {
					rs.allocated = true;
					const bool interior = in.dev && ((uintptr_t)(in.dev + cc->offset) & 15) == 0 && cc->offset + cc->size < in.n;
					if (interior)
						cc->d_in = in.dev + cc->offset; // interior chunk of a resident buffer: readable past its end
					else if (in.dev_chunks && !in.dev_chunks[cc->index] && cc->size)
						rs.rc = LRZGPU_E_PARAM; // a chunk this run was asked for but not given
					else if (!cc->in_buf.alloc((size_t)cc->size + 256, P.device))
						rs.rc = LRZGPU_E_NOMEM;
					else
						cc->d_in = cc->in_buf.p;
				}
				rc = rs.rc;
			}
			hipError_t e = hipSuccess;
			if (!rc && cc->in_buf.p) {
				if (!pieces) {
					// (a chunk handed over on its own has no readable bytes behind its end: it is copied next to padding)
					const uint8_t *from = in.dev ? in.dev + cc->offset : in.dev_chunks[cc->index];
					if (cc->size)
						e = hipMemcpyAsync(cc->in_buf.p, from, (size_t)cc->size, hipMemcpyDeviceToDevice, s);
				} else {
					// copy / pread of this reader's next piece under the DMA of its last one
					for (int64_t o = (int64_t)t * (int64_t)STAGE_BYTES; o < cc->size && !rc; o += (int64_t)n_readers * (int64_t)STAGE_BYTES, k ^= 1) {
						const size_t len = (size_t)(cc->size - o < (int64_t)STAGE_BYTES ? cc->size - o : (int64_t)STAGE_BYTES);
						if (used[k] && event_wait(done[k]) != hipSuccess) {
							rc = LRZGPU_E_HIP;
							break;
						}
						if (in.host)
							memcpy(stage[k], in.host + cc->offset + o, len);
						else if (pread_all(in.fd, stage[k], len, in.fd_base + cc->offset + o) != 0) {
							rc = LRZGPU_E_IO;
							break;
						}
						if (hipMemcpyAsync(cc->in_buf.p + o, stage[k], len, hipMemcpyHostToDevice, s) != hipSuccess ||
						    hipEventRecord(done[k], s) != hipSuccess)
							rc = LRZGPU_E_HIP;
						used[k] = true;
					}
				}
				if (!rc && e == hipSuccess && t == 0)
					e = hipMemsetAsync(cc->in_buf.p + cc->size, 0, 256, s);
				if (!rc && (e != hipSuccess || stream_wait(s) != hipSuccess))
					rc = LRZGPU_E_HIP;
			}
			if (rc) {
				fail(rc);
				break;
			}
			std::lock_guard<std::mutex> lk(mu);
			if (++read_state[m].arrived == n_readers) {
				cc->input_ready = true;
				cv.notify_all();
				if (tracing())
					fprintf(stderr, "lrzgpu reader: chunk %d (%lld bytes) in HBM at %.3f s\n", cc->index, (long long)cc->size, now_s() - t0);
			}
		}
		cleanup();
	}

	// ---- whole-input hash (the reference feeds it from cksumthread, src/rzip.c:564-584): MD5 unless
	// control->hash_code names another of hashes[] (src/main.c:64-79) ---------------------------------------
	uint8_t digest[64] = {0};
	const int hash_code = ctl->hash_code; // control->hash_code, src/rzip.c:943-950, 1195-1219
	// the hash of bytes that are in HBM: down in pinned pieces, piece k + 1 on its way while piece k is hashed (no host
	// CPU but the hashing itself: the DMA engine moves them)
	struct DeviceHashFeed {
		static constexpr size_t kPiece = (size_t)32 << 20;
		RawBuf<uint8_t> stage[2];
		hipStream_t s = nullptr;
		size_t pending = 0; // bytes of the piece on its way (in stage[k ^ 1] once waited for)
		int k = 0;
		double t_hash = 0, t_wait = 0;
		int open(int device)
		{
			if (hipSetDevice(device) != hipSuccess)
				return LRZGPU_E_HIP;
			stage[0].alloc(kPiece, true);
			stage[1].alloc(kPiece, true);
			return make_stream(&s) == hipSuccess ? 0 : LRZGPU_E_HIP;
		}
		// hashes what was on its way, after asking for the next piece (d == nullptr: nothing more to ask for)
		int step(Hasher &m, const uint8_t *d, size_t len)
		{
			if (pending && stream_wait_timed() != 0)
				return LRZGPU_E_HIP;
			const size_t have = pending;
			const int from = k;
			pending = 0;
			if (d && len) {
				k ^= 1;
				if (hipMemcpyAsync(stage[k].data(), d, len, hipMemcpyDeviceToHost, s) != hipSuccess)
					return LRZGPU_E_HIP;
				pending = len;
			}
			if (have) {
				const double ta = now_s();
				m.update(stage[from].data(), have);
				t_hash += now_s() - ta;
			}
			return 0;
		}
		int stream_wait_timed()
		{
			const double ta = now_s();
			const hipError_t e = stream_wait(s);
			t_wait += now_s() - ta;
			return e == hipSuccess ? 0 : -1;
		}
		int range(Hasher &m, const uint8_t *d, int64_t n, const std::atomic<int> &err)
		{
			for (int64_t o = 0; o < n && !err.load(); o += (int64_t)kPiece) {
				const int rc = step(m, d + o, (size_t)(n - o < (int64_t)kPiece ? n - o : (int64_t)kPiece));
				if (rc)
					return rc;
			}
			return 0;
		}
		int drain(Hasher &m) { return step(m, nullptr, 0); }
		void close()
		{
			if (s) {
				(void)stream_wait(s);
				StreamPool::get().give(s);
				s = nullptr;
			}
		}
	};
	// the hash of a range of the input file where the page cache holds it, through a mapping that moves along the file;
	// what cannot be mapped is read
	int hash_file_range(Hasher &m, int64_t from, int64_t n)
	{
		const size_t window = (size_t)256 << 20, piece = (size_t)32 << 20;
		const long pg = sysconf(_SC_PAGESIZE);
		std::vector<uint8_t> buf;
		for (int64_t o = 0; o < n && !P.error();) {
			const size_t len = (size_t)(n - o < (int64_t)window ? n - o : (int64_t)window);
			const int64_t file_off = in.fd_base + from + o, aligned = file_off / pg * pg;
			const size_t lead = (size_t)(file_off - aligned);
			void *mp = mmap(nullptr, len + lead, PROT_READ, MAP_SHARED, in.fd, (off_t)aligned);
			if (mp != MAP_FAILED) {
				(void)madvise(mp, len + lead, MADV_SEQUENTIAL);
				for (size_t q = 0; q < len && !P.error(); q += piece)
					m.update((const uint8_t *)mp + lead + q, len - q < piece ? len - q : piece);
				munmap(mp, len + lead);
			} else {
				buf.resize(piece);
				for (size_t q = 0; q < len && !P.error(); q += piece) {
					const size_t l2 = len - q < piece ? len - q : piece;
					if (pread_all(in.fd, buf.data(), l2, file_off + (int64_t)q) != 0)
						return LRZGPU_E_IO;
					m.update(buf.data(), l2);
				}
			}
			o += (int64_t)len;
		}
		return 0;
	}
	void md5_main()
	{
		std::unique_ptr<Hasher> hasher = make_hasher(hash_code);
		if (!hasher) {
			fail(LRZGPU_E_PARAM);
			return;
		}
		Hasher &m = *hasher;
		int rc = 0;
		if (in.host) {
			m.update(in.host, (size_t)in.n);
		} else if (in.n) {
			DeviceHashFeed feed;
			bool feed_open = false;
			if (in.dev) {
				rc = feed.open(P.device);
				feed_open = true;
				if (!rc)
					rc = feed.range(m, in.dev, in.n, P.err);
			} else {
				// a file.  The readers bring every chunk of this run into HBM for its scan: the hash takes it from there
				// (the chunk's copy stays until the hash has passed it), like an input that was in HBM from the start --
				// hashing out of the page cache, mapped or read, costs this thread the page faults or the memcpy of the
				// whole input, and this thread's speed is a floor of the run.  Chunks of the file that are not this
				// run's are hashed from the file.
				for (size_t c = 0; c < chunks.size() && !rc && !P.error(); c++) {
					ChunkCtx *cc = chunks[c].get();
					if (!cc->hash_holds) {
						if (feed_open)
							rc = feed.drain(m);
						if (!rc)
							rc = hash_file_range(m, cc->offset, cc->size);
						continue;
					}
					{
						std::unique_lock<std::mutex> lk(mu);
						cv.wait(lk, [&] { return P.err || cc->input_ready; });
						if (P.err)
							break;
					}
					if (!feed_open) {
						rc = feed.open(P.device);
						feed_open = true;
					}
					if (!rc)
						rc = feed.range(m, cc->d_in, cc->size, P.err);
					if (!rc)
						rc = feed.drain(m); // (the chunk's last piece is on the host before the copy may go)
					std::lock_guard<std::mutex> lk(mu);
					cc->hash_holds = false;
					if (cc->release_wanted) {
						cc->in_buf.release();
						cc->d_in = nullptr;
					}
				}
			}
			if (feed_open) {
				if (!rc)
					rc = feed.drain(m);
				if (tracing())
					fprintf(stderr, "lrzgpu hash thread: %.2f s hashing, %.2f s waiting for the next piece from the device; done at %.2f s\n", feed.t_hash,
						feed.t_wait, now_s() - t0);
				feed.close();
			}
			if (rc) {
				fail(rc);
				return;
			}
		}
		m.finish(digest);
	}

	// ---- one chunk through K1..K5 with early block release ---------------------------------------
	struct Scanner {
		Feeder F;
		ScanWorkspace *sw = nullptr;
		DevBuf runs;
		explicit Scanner(Pipeline &p) : F(p) {}
	};

	// Resolvers of different chunks must not share a CU: each is ONE latency-bound wavefront, and the dispatcher
	// happily packs eight single-wave workgroups onto the same SIMDs (measured: the first scan segment takes
	// 653 ms with eight resolvers side by side against 413 ms alone, with nothing else on the GPU).  Scanner k gets
	// the CUs k, k + 8, k + 16, ... for its scan stream: disjoint sets whatever the mask-bit-to-XCD mapping is (one
	// XCD each if the bits go round the XCDs), 32 CUs wide so that the K1 kernels on the same stream keep their
	// bandwidth.  Such streams are blocking streams: nothing in the pipeline uses the null stream.
	std::atomic<int> scanner_ids{0};
	hipError_t make_scan_stream(hipStream_t *s)
	{
		hipDeviceProp_t prop;
		if (scan_slots > 1 && hipGetDeviceProperties(&prop, P.device) == hipSuccess && prop.multiProcessorCount >= 64) {
			const int ncu = prop.multiProcessorCount > 256 ? 256 : prop.multiProcessorCount;
			const int k = scanner_ids.fetch_add(1) % 8;
			const int kind = 16 + k;
			if ((*s = StreamPool::get().take(P.device, kind)) != nullptr)
				return hipSuccess;
			uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
			for (int c = k; c < ncu; c += 8)
				mask[c >> 5] |= 1u << (c & 31);
			if (hipExtStreamCreateWithCUMask(s, (uint32_t)((ncu + 31) / 32), mask) == hipSuccess) {
				StreamPool::get().created(*s, P.device, kind);
				return hipSuccess;
			}
			(void)hipGetLastError();
		}
		return make_stream(s, true);
	}

	int scanner_open(Scanner &S)
	{
		if (hipSetDevice(P.device) != hipSuccess || make_scan_stream(&S.F.ms) != hipSuccess)
			return LRZGPU_E_HIP;
		const int ngate = scan_slots > 1 ? 2 : 6;
		for (int k = 0; k < ngate; k++) {
			hipStream_t gs;
			if (make_stream(&gs) != hipSuccess)
				return LRZGPU_E_HIP;
			S.F.gate_streams.push_back(gs);
		}
		const int64_t cap_chunk = P.sz.max_chunk < in.n ? P.sz.max_chunk : in.n;
		S.sw = WorkspacePool::get().take_scan(P.sz.rzip_level, cap_chunk, P.device);
		return S.sw ? 0 : LRZGPU_E_NOMEM;
	}
	void scanner_close(Scanner &S)
	{
		S.F.destroy();
		const int64_t cap_chunk = P.sz.max_chunk < in.n ? P.sz.max_chunk : in.n;
		WorkspacePool::get().give_scan(S.sw, P.sz.rzip_level, cap_chunk, P.device);
		S.sw = nullptr;
		S.runs.release();
	}

	// gathers stream-1 bytes [S0, S1) from `runs` (absolute dst offsets)
	int gather(Scanner &S, ChunkCtx *cc, const std::vector<CopyRun> &runs, int64_t S0, int64_t S1)
	{
		if (runs.empty() || S1 <= S0)
			return 0;
		hipStream_t ms = S.F.ms;
		if (runs.size() * sizeof(CopyRun) > S.runs.cap && !S.runs.alloc((runs.size() * 2 + 64) * sizeof(CopyRun), P.device))
			return LRZGPU_E_NOMEM;
		if (hipMemcpyAsync(S.runs.p, runs.data(), runs.size() * sizeof(CopyRun), hipMemcpyHostToDevice, ms) != hipSuccess)
			return LRZGPU_E_HIP;
		EventTimer tg(ms);
		int gr = gather_runs_device(cc->d_in, cc->stream1.p, (const CopyRun *)S.runs.p, (int)runs.size(), S0, S1, ms);
		tg.stop();
		if (gr != 0 || stream_wait(ms) != hipSuccess)
			return LRZGPU_E_HIP;
		ProfileStore &ps = ProfileStore::get();
		std::lock_guard<std::mutex> lk(ps.mu);
		ps.p.gather_ms += tg.ms_noted(ps, PK_GATHER);
		ps.p.gather_launches++;
		ps.p.gather_bytes += S1 - S0;
		return 0;
	}

	static bool census_allowed()
	{
		const char *e = getenv("LRZGPU_CENSUS"); // 0: every chunk through the resolver (tests compare both ways)
		return !(e && *e == '0');
	}
	int scan_chunk(Scanner &S, ChunkCtx *cc, int64_t vr_in)
	{
		const int64_t chunk_size = cc->size, bufsize = P.sz.stream_bufsize;
		Feeder &F = S.F;
		cc->vr_in = vr_in;
		cc->file_order.clear();
		cc->stream0.clear();
		if (!cc->stream1.p && !cc->stream1.alloc((size_t)chunk_size + 256, P.device))
			return LRZGPU_E_NOMEM;
		{
			int pr = F.poll(true); // nothing of an earlier chunk may still sit in the descriptor arena
			if (pr)
				return pr;
			int rr = F.reserve((size_t)(chunk_size / bufsize + 8) * 2);
			if (rr)
				return rr;
		}
		// ---- speculative early emission while the scan runs -----------------------------------
		int64_t E = 0;          // chunk position up to which stream-1 bytes have been gathered
		int64_t Sg = 0;         // stream-1 bytes gathered so far
		int64_t seen = 0;       // match records consumed
		int64_t blocks_out = 0; // full stream-1 blocks already submitted
		bool violated = false;
		std::map<int64_t, Job *> early; // stream-1 offset -> job
		std::set<Job *> starting;       // early jobs whose block is not complete yet
		std::vector<MatchRec> rec_buf;
		// early start of blocks: LZMA blocks only, and not under a filter (a block is filtered once, in place, whole)
		const bool can_early = speculate && P.early_mode != 0 && !P.sz.zstd && !P.sz.no_compress && !P.filter_flag && bufsize >= 4096;
		auto make_early = [&](Job *j) {
			std::lock_guard<std::mutex> lk(P.mu);
			j->early = true;
			P.early_unclaimed++;
			P.n_early_jobs++;
		};

		auto advance = [&](const ScanState &h, int64_t upto, bool final_call, const std::vector<MatchRec> *final_recs) -> int {
			const int64_t nrec = final_recs ? (int64_t)final_recs->size() : h.n_records;
			std::vector<CopyRun> runs;
			const int64_t S_before = Sg;
			if (nrec > seen) {
				const MatchRec *rp;
				if (final_recs)
					rp = final_recs->data() + seen;
				else {
					rec_buf.resize((size_t)(nrec - seen));
					if (d2h_pageable(rec_buf.data(), S.sw->records + seen, (size_t)(nrec - seen) * sizeof(MatchRec), F.ms) != hipSuccess)
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
						runs.push_back(CopyRun{E, Sg, r.p - E});
						Sg += r.p - E;
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
					runs.push_back(CopyRun{E, Sg, Fp - E});
				Sg += Fp - E;
				E = Fp;
			}
			if (Sg > S_before) {
				int g = gather(S, cc, runs, S_before, Sg);
				if (g)
					return g;
			}
			if (final_call)
				return 0;
			std::vector<Job *> fresh;
			while ((blocks_out + 1) * bufsize <= Sg) {
				const int64_t off = blocks_out * bufsize;
				auto it = early.find(off);
				Job *j = it != early.end() ? it->second : nullptr; // started while it was being filled: now whole
				if (j)
					starting.erase(j);
				else {
					j = F.new_job(cc, BlockRef{1, off, bufsize});
					// encoders with nothing to do: the finder hands them a first part of the block at once
					if (can_early && P.early_split && P.want_early())
						make_early(j);
					early[off] = j;
				}
				fresh.push_back(j);
				blocks_out++;
			}
			{
				std::lock_guard<std::mutex> lk(mu);
				n_early += (int64_t)fresh.size();
			}
			int sr2 = F.submit(fresh);
			if (sr2)
				return sr2;
			// the block under construction: once a first part of it is there and encoders have
			// nothing to do, the finder runs on what is there and an encoder starts on those lists; every further
			// piece is another run on the longer prefix
			if (can_early) {
				const int64_t off = blocks_out * bufsize, have = Sg - off;
				auto it = early.find(off);
				Job *ej = it != early.end() ? it->second : nullptr;
				if (!ej && have >= P.early_first && P.want_early()) {
					ej = F.new_job(cc, BlockRef{1, off, bufsize});
					make_early(ej);
					early[off] = ej;
					starting.insert(ej);
				}
				if (ej)
					P.stage(ej, have);
			}
			if (P.error())
				return P.error();
			return F.poll(false);
		};

		ScanProgressFn progress = nullptr;
		if (speculate)
			progress = [&](const ScanState &h, int64_t upto) -> int { return advance(h, upto, false, nullptr); };

		ScanResult sr;
		int64_t vr = vr_in;
		// (the file's last chunk: nobody needs the victim_round it ends with, so it may skip the resolver when it holds
		// no repeat at all -- rzip_census.hip)
		int r = scan_chunk_device(S.sw, cc->d_in, chunk_size, P.sz.rzip_level, &vr, &sr, F.ms, progress, cc->last && census_allowed());
		if (r)
			return r < -50 ? r : (r == -4 ? LRZGPU_E_NOMEM : LRZGPU_E_INTERNAL);
		{
			std::lock_guard<std::mutex> lk(mu); // scanners read a predecessor's vr_out under mu (predicted_vr)
			cc->vr_out = vr;
		}
		EmitResult er;
		emit_streams(sr.records, chunk_size, cc->chunk_bytes, sr.crc, &er);
		cc->stream0.swap(er.stream0);
		cc->stream1_len = er.stream1_len;
		if (speculate && !violated) {
			int a = advance(sr.final_state, chunk_size, true, &sr.records);
			if (a)
				return a;
		}
		if (!speculate || violated || Sg != er.stream1_len) {
			// (re)build stream 1 from the final run table; early blocks, if any, are void
			if (speculate && violated) {
				ProfileStore &ps = ProfileStore::get();
				std::lock_guard<std::mutex> lk(ps.mu);
				ps.p.spec_rollbacks++;
			}
			if (!early.empty()) {
				{
					std::lock_guard<std::mutex> lk(mu);
					n_violations++;
				}
				{
					ProfileStore &ps = ProfileStore::get();
					std::lock_guard<std::mutex> lk(ps.mu);
					ps.p.spec_cancelled_blocks += (int64_t)early.size();
				}
				std::vector<Job *> dead;
				for (auto &kv : early)
					dead.push_back(kv.second);
				for (Job *j : dead)
					j->cancelled = true;
				int pr = F.poll(true);
				if (pr)
					return pr;
				P.cancel_and_wait(dead);
				if (P.error())
					return P.error();
				early.clear();
			}
			int g = gather(S, cc, er.runs, 0, er.stream1_len);
			if (g)
				return g;
		}
		if (hipMemsetAsync(cc->stream1.p + er.stream1_len, 0, 256, F.ms) != hipSuccess || stream_wait(F.ms) != hipSuccess)
			return LRZGPU_E_HIP;
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
					if (starting.erase(j))
						fresh.push_back(j); // started early, completed by the last piece of the scan: the whole block now
				}
			}
			if (!j) {
				j = F.new_job(cc, br);
				fresh.push_back(j);
			}
			cc->file_order.push_back(j);
		}
		if (!early.empty()) {
			// a block started early that the chunk's last, shorter block took the place of: withdrawn (every other
			// early block is a full stream-1 block of the final layout)
			std::vector<Job *> dead;
			for (auto &kv : early) {
				if (!starting.count(kv.second))
					return LRZGPU_E_INTERNAL;
				dead.push_back(kv.second);
			}
			P.cancel_and_wait(dead);
			if (P.error())
				return P.error();
			early.clear();
		}
		return F.submit(fresh);
	}

	void scanner_main()
	{
		Scanner S(P);
		int rc = scanner_open(S);
		while (!rc) {
			ChunkCtx *cc = nullptr;
			int64_t vr_in = 0;
			{
				std::unique_lock<std::mutex> lk(mu);
				if (P.err || next_scan >= mine.size())
					break;
				const size_t m = next_scan++;
				cc = chunks[(size_t)mine[m]].get();
				cv.wait(lk, [&] { return P.err || cc->input_ready; });
				if (P.err)
					break;
				vr_in = predicted_vr(cc->index);
			}
			rc = scan_chunk(S, cc, vr_in);
			if (!rc)
				rc = S.F.poll(true);
			if (rc)
				break;
			std::lock_guard<std::mutex> lk(mu);
			cc->scanned = true;
			cc->t_scanned = now_s();
			cv.notify_all();
		}
		if (rc)
			fail(rc);
		scanner_close(S);
	}

	// victim_round a chunk should start from (mu held): what its predecessor left if that is known, the
	// caller's hint for the first chunk of a partial run, else 0 -- the value only moves when a tag value
	// collects max_chain_len table entries, which ordinary data does rarely
	int64_t predicted_vr(int index)
	{
		if (index == 0)
			return 0;
		if (sel && sel->victim_in && sel->victim_in[index] >= 0)
			return sel->victim_in[index]; // the caller's word comes first (lrzgpu.h: "gives the value chunk k starts from")
		const ChunkCtx *prev = chunks[(size_t)index - 1].get();
		if (prev->scanned)
			return prev->vr_out;
		return 0;
	}

	int run();
};

int Run::run()
{
	int rc = select_device(ctl->device);
	if (rc)
		return rc;
	P.ctl = ctl;
	P.device = ctl->device;
	rc = sizing_for_input(ctl, in.n, &P.sz);
	if (rc)
		return rc;
	if (control_filter(ctl, &P.filter_flag, &P.filter_delta) || hash_length(hash_code) < 0)
		return LRZGPU_E_PARAM;
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
	speculate = true; // (blocks are released to the back end while their chunk is still being scanned)
	// early start of blocks.  LRZGPU_EARLY_START: 0 off, 1 (default) while encoders have nothing to
	// do, 2 every block (tests); LRZGPU_EARLY_STEP: bytes of a block between two finder runs (default 1/16 of a block).
	// None of it changes the output.
	{
		const char *e = getenv("LRZGPU_EARLY_START"); // read per run: tests flip it inside one process
		P.early_mode = e ? atoi(e) : 1;
		if (P.early_mode < 0 || P.early_mode > 2)
			P.early_mode = 1;
		int64_t step = P.sz.stream_bufsize / 16;
		if (const char *t = getenv("LRZGPU_EARLY_STEP"))
			if (atoll(t) > 0)
				step = atoll(t);
		if (step < 4096)
			step = 4096;
		P.early_step = step;
		P.early_first = step;
		P.early_split = true;
	}

	// the chunks of the file (src/rzip.c:1041: at least one pass, even for an empty input; STDIN mode: one more,
	// empty, when the input ends exactly where a chunk does)
	{
		std::vector<int64_t> sizes;
		chunk_sizes_for(ctl, P.sz, in.n, &sizes);
		int64_t offset = 0;
		for (size_t k = 0; k < sizes.size(); k++) {
			std::unique_ptr<ChunkCtx> cc(new ChunkCtx());
			cc->index = (int)k;
			cc->offset = offset;
			cc->size = sizes[k];
			cc->chunk_bytes = chunk_bytes_for(cc->size);
			offset += cc->size;
			cc->last = k + 1 == sizes.size();
			chunks.push_back(std::move(cc));
		}
	}
	for (size_t k = 0; k < chunks.size(); k++)
		if (!sel || (sel->stride > 0 && (int)k % sel->stride == sel->first))
			mine.push_back((int)k);
	scan_slots = ctl->scan_slots > 0 ? ctl->scan_slots : 8;
	if ((size_t)scan_slots > mine.size())
		scan_slots = mine.empty() ? 1 : (int)mine.size();
	// Every GPU worker owns a finder workspace of ~240 bytes per byte of the largest block, for the whole run: with the
	// 134 MB blocks of a 32 GiB chunk that is 32 GB each, and eight of them beside the chunk (input copy + stream 1) do
	// not fit 288 GB.  So: as many workers as fit what the device has left beside the chunks in flight (at least one; the
	// output does not depend on the number).  Parked pool memory counts as free (it is given back on demand).
	if (!P.sz.zstd && !P.sz.no_compress && in.n > 0) {
		size_t in_flight = 0;
		for (int k : mine) {
			const size_t c = (size_t)chunks[(size_t)k]->size;
			if (c > in_flight)
				in_flight = c;
		}
		// per scanner: stream 1 of the chunk, a copy of it unless the caller's device buffer can be used in place, ~4 GB
		// of scan workspace
		in_flight = (size_t)(scan_slots + (in.dev ? 0 : 1)) * (2 * in_flight + ((size_t)4 << 30));
		const size_t avail = DeviceBudget::free_now() + WorkspacePool::get().idle_bytes + DevicePool::get().idle_bytes;
		size_t per_ws = WorkspacePool::mf_bytes((size_t)P.sz.stream_bufsize, P.mf_per_pos);
		if (avail != ~(size_t)0 && per_ws > ((size_t)1 << 30)) { // (small blocks: nothing to bound)
			const size_t room = avail > in_flight + DeviceBudget::margin() ? avail - in_flight - DeviceBudget::margin() : 0;
			// One workspace must fit.  Its list pools are sized for 16 entries per block byte (text needs ~5, and a
			// finder run that outgrows its pool is repeated with a larger one): a block too large for that gets what
			// fits, down to 4 entries per byte; below that the block is beyond this device -- said now, not as an
			// out-of-memory error minutes into the run (the ceiling: lrzgpu_max_block_bytes()).
			while (per_ws > room && P.mf_per_pos > kMinPoolPerPos) {
				P.mf_per_pos = P.mf_per_pos > 8 ? P.mf_per_pos - 4 : P.mf_per_pos - 2;
				per_ws = WorkspacePool::mf_bytes((size_t)P.sz.stream_bufsize, P.mf_per_pos);
			}
			if (per_ws > room) {
				if (ctl->verbose || getenv("LRZGPU_TRACE"))
					fprintf(stderr, "lrzgpu: blocks of %lld bytes need a match-finder workspace of %zu MiB; %zu MiB are free beside the chunks in flight\n",
						(long long)P.sz.stream_bufsize, per_ws >> 20, room >> 20);
				return LRZGPU_E_BLOCK_TOO_LARGE;
			}
			size_t fit = room / per_ws;
			if (fit < 1)
				fit = 1;
			if ((size_t)P.n_gpu_workers > fit) {
				if (getenv("LRZGPU_TRACE"))
					fprintf(stderr, "lrzgpu driver: %d GPU workers asked for, %zu finder workspaces of %zu MiB fit beside the chunks in flight\n",
						P.n_gpu_workers, fit, per_ws >> 20);
				P.n_gpu_workers = (int)fit;
				P.held_limit = (size_t)P.n_encoders + (size_t)P.n_gpu_workers + 2;
			}
		}
	}
	if (ctl->verbose)
		fprintf(stderr, "lrzgpu: threads %d bufsize %lld dict %u chunk %lld chunks %zu (%zu here) scanners %d encoders %d gpu workers %d\n",
			P.sz.threads, (long long)P.sz.stream_bufsize, P.sz.dict_size, (long long)P.sz.max_chunk, chunks.size(), mine.size(),
			scan_slots, P.n_encoders, P.n_gpu_workers);

	t0 = now_s();
	g_trace_t0 = t0;
	g_trace_events.store((getenv("LRZGPU_TRACE") && atoi(getenv("LRZGPU_TRACE")) >= 2) ? 1 : 0, std::memory_order_relaxed);
	P.on_fail = [this] {
		std::lock_guard<std::mutex> lk(mu);
		cv.notify_all();
	};
	const bool want_md5 = !sel || sel->with_md5;
	if (in.dev_chunks && (want_md5 || !sel))
		return LRZGPU_E_PARAM; // the whole-input hash needs the whole input (checked before any thread exists)
	if (want_md5 && !in.host && !in.dev && !in.dev_chunks)
		for (int k : mine)
			chunks[(size_t)k]->hash_holds = true; // a file: the hash reads this run's chunks from their copies in HBM (md5_main)
	P.start();
	std::vector<std::thread> side;
	if (want_md5)
		side.emplace_back([this] { P.guarded([this] { md5_main(); }, 3); });
	n_readers = (in.dev || in.dev_chunks) ? 1 : std::max(1, std::min(scan_slots, 8));
	read_state.assign(mine.size(), ReadState());
	for (int t = 0; t < n_readers; t++)
		side.emplace_back([this, t] { P.guarded([this, t] { reader_main(t); }, 4); });
	for (int k = 0; k < scan_slots; k++)
		side.emplace_back([this] { P.guarded([this] { scanner_main(); }, 2); });

	// ---- committer: chunks in file order ------------------------------------------------------------
	int ret = 0;
	std::unique_ptr<Scanner> rescanner;
	double t_scan_last = 0, t_blocks = 0;
	const bool whole_file = !sel;
	if (whole_file && out.begin(21) != 0) // magic placeholder (compress_file, src/lrzip.c:1487-1547)
		ret = LRZGPU_E_IO;
	for (size_t m = 0; m < mine.size() && !ret; m++) {
		ChunkCtx *cc = chunks[(size_t)mine[m]].get();
		{
			std::unique_lock<std::mutex> lk(mu);
			cv.wait(lk, [&] { return P.err || cc->scanned; });
			if (P.err) {
				ret = P.err;
				break;
			}
		}
		// victim_round chain: only checkable when this run also scanned the predecessor
		if (cc->index > 0 && (!sel || sel->stride == 1)) {
			const ChunkCtx *prev = chunks[(size_t)cc->index - 1].get();
			if (prev->vr_out != cc->vr_in) {
				// the chunk was scanned from the wrong victim_round: void its blocks and scan it again
				n_rescans++;
				std::vector<Job *> dead;
				for (auto &j : cc->jobs)
					dead.push_back(j.get());
				P.cancel_and_wait(dead);
				if (P.error()) {
					ret = P.error();
					break;
				}
				if (!rescanner) {
					rescanner.reset(new Scanner(P));
					int orc = scanner_open(*rescanner);
					if (orc) {
						ret = orc;
						break;
					}
				}
				{
					std::lock_guard<std::mutex> lk(mu);
					cc->scanned = false; // its vr_out is not to be trusted while it is scanned again
				}
				int src = scan_chunk(*rescanner, cc, prev->vr_out);
				if (!src)
					src = rescanner->F.poll(true);
				if (src) {
					ret = src;
					break;
				}
				{
					std::lock_guard<std::mutex> lk(mu);
					cc->scanned = true;
					cv.notify_all();
				}
			}
		}
		t_scan_last = cc->t_scanned;
		{
			// no rescan can be asked for any more: the input copy goes (now, or when the hash has passed it)
			std::lock_guard<std::mutex> lk(mu);
			if (cc->hash_holds)
				cc->release_wanted = true;
			else {
				cc->in_buf.release();
				cc->d_in = nullptr;
			}
		}
		// wait for every block of the chunk (discarded early ones included: they reference its buffers)
		{
			std::unique_lock<std::mutex> lk(P.mu);
			P.cv_done.wait(lk, [&] {
				if (P.err)
					return true;
				for (auto &j : cc->jobs)
					if (!j->finished)
						return false;
				return true;
			});
			if (P.err) {
				ret = P.err;
				break;
			}
		}
		t_blocks = now_s();
		cc->stream1.release();
		// ordered container assembly of this chunk, straight into the sink's memory where it offers some
		std::vector<DoneBlock> blocks;
		for (Job *j : cc->file_order)
			blocks.push_back(std::move(j->done));
		const size_t total = chunk_image_size(cc->chunk_bytes, blocks);
		uint8_t *space = (sel && sel->on_chunk) ? nullptr : out.append_space(total);
		if (space) {
			write_chunk_raw(space, cc->chunk_bytes, cc->last, cc->size, blocks);
		} else {
			std::unique_ptr<uint8_t, void (*)(void *)> img((uint8_t *)malloc(total ? total : 1), free);
			if (!img)
				ret = LRZGPU_E_NOMEM;
			else {
				write_chunk_raw(img.get(), cc->chunk_bytes, cc->last, cc->size, blocks);
				if (sel && sel->on_chunk) {
					if (sel->on_chunk(sel->ctx, cc->index, cc->vr_in, cc->vr_out, img.get(), (int64_t)total) != 0)
						ret = LRZGPU_E_IO;
				} else if (out.put(img.get(), total) != 0)
					ret = LRZGPU_E_IO;
			}
		}
		cc->jobs.clear();
		cc->file_order.clear();
		std::vector<uint8_t>().swap(cc->stream0);
		std::lock_guard<std::mutex> lk(mu);
		committed = m + 1;
		cv.notify_all();
	}
	if (ret)
		fail(ret);
	if (rescanner)
		scanner_close(*rescanner);
	// every thread ends on its own (work done) or on the error flag; chunks and jobs outlive them all
	for (auto &t : side)
		t.join();
	{
		// blocks still in flight after a failure reference chunk buffers: wait them out before those go
		std::unique_lock<std::mutex> lk(P.mu);
		if (!P.err)
			P.cv_done.wait(lk, [&] {
				for (auto &c : chunks)
					for (auto &j : c->jobs)
						if (!j->finished)
							return false;
				return true;
			});
	}
	P.stop();
	const double t_md5 = now_s();
	if (!ret && P.err)
		ret = P.err;
	if (ret)
		return ret;

	if (whole_file) {
		// the hash after the last chunk (none for code 0, "CRC": the chunk CRCs are all there is) and its code in magic[14]
		const int hash_len = hash_code == 0 ? 0 : hash_length(hash_code);
		if (hash_len > 0 && out.put(digest, (size_t)hash_len) != 0)
			return LRZGPU_E_IO;
		uint8_t magic[21];
		write_magic_for(magic, ctl, P.sz, in.n, chunks.size());
		if (out.finish(magic, 21) != 0)
			return LRZGPU_E_IO;
	}
	if (want_md5) {
		memcpy(ctl->hash_resblock, digest, 16);
		memcpy(ctl->hash_full, digest, sizeof(ctl->hash_full));
	}
	if (tracing())
		fprintf(stderr, "lrzgpu driver: %zu chunks, %d scanners: last scan done %.2f  last finder %.2f  last encode %.2f  all blocks %.2f  md5 joined %.2f  assembled %.2f s (since start); early blocks %lld, redone chunks %lld, rescans (victim_round) %lld; worker sums: block copy+gate %.2f finder %.2f lists D2H %.2f, encoders busy %.2f idle %.2f s\n",
			chunks.size(), scan_slots, t_scan_last - t0, P.t_last_mf - t0, P.t_last_enc - t0, t_blocks - t0, t_md5 - t0, now_s() - t0,
			(long long)n_early, (long long)n_violations, (long long)n_rescans, P.blk_busy, P.mf_busy, P.d2h_busy, P.enc_busy, P.enc_wait);
	{
		ProfileStore &ps = ProfileStore::get();
		std::lock_guard<std::mutex> lk(ps.mu);
		ps.p.victim_rescans += n_rescans;
		const double vals[8] = {P.enc_busy, P.enc_wait, P.mf_busy, P.d2h_busy, t_scan_last - t0, P.t_last_mf - t0, P.t_last_enc - t0, now_s() - t0};
		for (int k = 0; k < 8; k++)
			ps.p.pipeline_s[k] += vals[k] > 0 ? vals[k] : 0;
		ps.p.early_s[0] += P.t_first_enc > t0 ? P.t_first_enc - t0 : 0;
		ps.p.early_s[1] += P.rest_wait;
		ps.p.early_s[2] += (double)P.n_early_jobs;
		ps.p.early_s[3] += (double)P.n_early_stages;
	}
	{
		LzmaParams p;
		if (!P.sz.no_compress && lzma_normalize(p, P.sz.level, P.sz.dict_size, 3, 0, 2, P.sz.level < 7 ? 32 : 64) == LZ_OK)
			lzma_write_props(p, ctl->lzma_properties);
	}
	return 0;
}

int run_compress(lrzgpu_control *ctl, const CompressSource &in, CompressSink &out, const ChunkSelect *sel)
{
	try {
		Run r(ctl, in, out, sel);
		return r.run();
	} catch (const std::bad_alloc &) {
		return LRZGPU_E_NOMEM;
	} catch (...) {
		return LRZGPU_E_INTERNAL;
	}
}

// ---- sinks ----------------------------------------------------------------------------------------
MemorySink::~MemorySink() { free(p); }
uint8_t *MemorySink::append_space(size_t n)
{
	if (len + n > cap) {
		size_t want = cap + cap / 2;
		if (want < len + n)
			want = len + n;
		if (want < ((size_t)1 << 20))
			want = (size_t)1 << 20;
		uint8_t *q = (uint8_t *)realloc(p, want);
		if (!q)
			return nullptr;
		p = q;
		cap = want;
	}
	uint8_t *r = p + len;
	len += n;
	return r;
}
int MemorySink::begin(size_t placeholder)
{
	len = 0;
	uint8_t *q = append_space(placeholder);
	if (!q)
		return -1;
	memset(q, 0, placeholder);
	return 0;
}
int MemorySink::put(const uint8_t *q, size_t n)
{
	uint8_t *d = append_space(n);
	if (!d)
		return -1;
	memcpy(d, q, n);
	return 0;
}
int MemorySink::finish(const uint8_t *head, size_t n)
{
	if (len < n)
		return -1;
	memcpy(p, head, n);
	return 0;
}

int FdSink::begin(size_t placeholder)
{
	// compress_file() reserves the magic up front and rewrites it at the end (src/lrzip.c:1487-1555);
	// rzip_fd() alone writes no magic at all
	if (!with_magic)
		return 0;
	start = lseek(fd, 0, SEEK_CUR);
	seekable = start >= 0;
	if (!seekable) {
		held.assign(placeholder, 0); // a pipe: the image is held back until the magic is known
		return 0;
	}
	std::vector<uint8_t> z(placeholder, 0);
	return write_all(fd, z.data(), z.size());
}
int FdSink::put(const uint8_t *p, size_t n)
{
	if (with_magic && !seekable) {
		held.insert(held.end(), p, p + n);
		return 0;
	}
	return write_all(fd, p, n);
}
int FdSink::finish(const uint8_t *head, size_t n)
{
	if (!with_magic)
		return 0;
	if (!seekable) {
		memcpy(held.data(), head, n);
		return write_all(fd, held.data(), held.size());
	}
	const off_t end = lseek(fd, 0, SEEK_CUR);
	if (end < 0 || lseek(fd, start, SEEK_SET) < 0 || write_all(fd, head, n) != 0 || lseek(fd, end, SEEK_SET) < 0)
		return -1;
	return 0;
}

} // namespace lrzgpu

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

template <typename F> static int abi_guard(F &&f) // nothing may leave an extern "C" entry point
{
	try {
		return f();
	} catch (const std::bad_alloc &) {
		return LRZGPU_E_NOMEM;
	} catch (...) {
		return LRZGPU_E_INTERNAL;
	}
}

static int compress_to_malloc(lrzgpu_control *control, const CompressSource &src, uint8_t **out, int64_t *out_len)
{
	MemorySink sink;
	int r = run_compress(control, src, sink, nullptr);
	if (r)
		return r;
	*out_len = (int64_t)sink.len;
	*out = sink.release(); // the image as the sink built it: no copy
	return 0;
}

extern "C" int lrzgpu_compress_buffer(lrzgpu_control *control, const uint8_t *in, int64_t n, uint8_t **out, int64_t *out_len)
{
	if (!control || !out || !out_len || n < 0 || (!in && n))
		return LRZGPU_E_PARAM;
	return abi_guard([&] {
		CompressSource s;
		static const uint8_t empty = 0;
		s.host = in ? in : &empty;
		s.n = n;
		return compress_to_malloc(control, s, out, out_len);
	});
}

extern "C" int lrzgpu_compress_buffer_dev(lrzgpu_control *control, const void *d_in, int64_t n, uint8_t **out, int64_t *out_len)
{
	if (!control || !out || !out_len || n < 0 || (!d_in && n))
		return LRZGPU_E_PARAM;
	return abi_guard([&] {
		CompressSource s;
		s.dev = (const uint8_t *)d_in;
		s.n = n;
		if (n == 0) {
			static const uint8_t empty = 0;
			s.dev = nullptr;
			s.host = &empty;
		}
		return compress_to_malloc(control, s, out, out_len);
	});
}

// fd_in -> source.  Regular files are read chunk by chunk as the scan needs them (the reference maps one
// chunk at a time, src/rzip.c:1057-1107).  With control->stdin_mode (FLAG_STDIN) the fd is read as a stream from
// its current offset, whatever it is, and the run chunks it the way mmap_stdin() does (src/rzip.c:800-836: the
// reference fills one anonymous mapping per chunk; the bytes are the same, so they are spooled here and cut up by
// chunk_sizes_for()).  A pipe without stdin_mode is spooled and compressed like a regular file of that size.
static int source_from_fd(int fd, bool as_stream, CompressSource *s, std::vector<uint8_t> *spool)
{
	const off_t cur = as_stream ? (off_t)-1 : lseek(fd, 0, SEEK_CUR);
	if (cur < 0) {
		if (!as_stream && errno != ESPIPE)
			return LRZGPU_E_IO;
		std::vector<uint8_t> tmp((size_t)1 << 20);
		for (;;) {
			ssize_t r = read(fd, tmp.data(), tmp.size());
			if (r < 0) {
				if (errno == EINTR)
					continue;
				return LRZGPU_E_IO;
			}
			if (r == 0)
				break;
			spool->insert(spool->end(), tmp.data(), tmp.data() + r);
		}
		static const uint8_t empty = 0;
		s->host = spool->empty() ? &empty : spool->data();
		s->n = (int64_t)spool->size();
		return 0;
	}
	const off_t end = lseek(fd, 0, SEEK_END);
	if (end < 0 || lseek(fd, 0, SEEK_SET) < 0)
		return LRZGPU_E_IO;
	s->fd = fd;
	s->fd_base = 0;
	s->n = (int64_t)end;
	if (s->n == 0) {
		static const uint8_t empty = 0;
		s->fd = -1;
		s->host = &empty;
	}
	return 0;
}

static int compress_fd(lrzgpu_control *control, int fd_in, int fd_out, bool with_magic)
{
	if (!control)
		return LRZGPU_E_PARAM;
	return abi_guard([&] {
		CompressSource s;
		std::vector<uint8_t> spool;
		int r = source_from_fd(fd_in, control->stdin_mode != 0, &s, &spool);
		if (r)
			return r;
		FdSink sink;
		sink.fd = fd_out;
		sink.with_magic = with_magic;
		return run_compress(control, s, sink, nullptr);
	});
}

extern "C" int lrzgpu_rzip_fd(lrzgpu_control *control, int fd_in, int fd_out) { return compress_fd(control, fd_in, fd_out, false); }

extern "C" int lrzgpu_compress_file(lrzgpu_control *control, int fd_in, int fd_out) { return compress_fd(control, fd_in, fd_out, true); }

// ---- chunk-sharded compression: one process per GPU, one file ------------------------------------
extern "C" int lrzgpu_compress_chunks_dev(lrzgpu_control *control, const void *d_in, int64_t n, int first, int stride,
					  const int64_t *victim_in, int with_md5, lrzgpu_chunk_fn on_chunk, void *ctx)
{
	if (!control || n < 0 || (!d_in && n) || stride < 1 || first < 0 || first >= stride || !on_chunk)
		return LRZGPU_E_PARAM;
	return abi_guard([&] {
		CompressSource s;
		s.dev = (const uint8_t *)d_in;
		s.n = n;
		if (n == 0) {
			static const uint8_t empty = 0;
			s.dev = nullptr;
			s.host = &empty;
		}
		ChunkSelect sel;
		sel.first = first;
		sel.stride = stride;
		sel.victim_in = victim_in;
		sel.with_md5 = with_md5 != 0;
		sel.on_chunk = on_chunk;
		sel.ctx = ctx;
		MemorySink unused;
		return run_compress(control, s, unused, &sel);
	});
}

extern "C" int lrzgpu_compress_chunks(lrzgpu_control *control, const uint8_t *in, int64_t n, int first, int stride,
				      const int64_t *victim_in, int with_md5, lrzgpu_chunk_fn on_chunk, void *ctx)
{
	if (!control || n < 0 || (!in && n) || stride < 1 || first < 0 || first >= stride || !on_chunk)
		return LRZGPU_E_PARAM;
	return abi_guard([&] {
		CompressSource s;
		static const uint8_t empty = 0;
		s.host = in ? in : &empty;
		s.n = n;
		ChunkSelect sel;
		sel.first = first;
		sel.stride = stride;
		sel.victim_in = victim_in;
		sel.with_md5 = with_md5 != 0;
		sel.on_chunk = on_chunk;
		sel.ctx = ctx;
		MemorySink unused;
		return run_compress(control, s, unused, &sel);
	});
}

// rank 0's half: the file from finished chunk images, in order (magic, chunks, MD5) -- host only
extern "C" int lrzgpu_assemble_chunks(lrzgpu_control *control, int64_t st_size, int n_chunks, const uint8_t *const *chunk_img,
				      const int64_t *chunk_len, const uint8_t *digest, uint8_t **out, int64_t *out_len)
{
	// digest: lrzgpu_hash_length(control->hash_code) bytes (up to 64); none read for hash code 0
	if (!control || st_size < 0 || n_chunks < 1 || !chunk_img || !chunk_len || (!digest && control->hash_code != 0) || !out || !out_len)
		return LRZGPU_E_PARAM;
	return abi_guard([&] {
		Sizing s;
		int r = sizing_for_input(control, st_size, &s);
		if (r)
			return r;
		int ff = 0, fd = 0;
		const int hash_len = control->hash_code == 0 ? 0 : hash_length(control->hash_code);
		if (hash_len < 0 || control_filter(control, &ff, &fd))
			return LRZGPU_E_PARAM;
		size_t total = 21 + (size_t)hash_len;
		for (int c = 0; c < n_chunks; c++) {
			if (chunk_len[c] < 0 || !chunk_img[c])
				return LRZGPU_E_PARAM;
			total += (size_t)chunk_len[c];
		}
		uint8_t *o = (uint8_t *)malloc(total);
		if (!o)
			return LRZGPU_E_NOMEM;
		write_magic_for(o, control, s, st_size, (size_t)n_chunks);
		size_t at = 21;
		for (int c = 0; c < n_chunks; c++) {
			big_copy(o + at, chunk_img[c], (size_t)chunk_len[c]);
			at += (size_t)chunk_len[c];
		}
		if (hash_len)
			memcpy(o + at, digest, (size_t)hash_len);
		*out = o;
		*out_len = (int64_t)total;
		control->st_size = st_size;
		control->stream_bufsize = s.stream_bufsize;
		control->dictSize_used = s.dict_size;
		control->threads_used = s.threads;
		return 0;
	});
}

// ---- host-only helpers (no device needed) ----------------------------------------------------

// Would malloc(bytes) fail on this host right now?  open_stream_out() probes exactly that (src/stream.c:1291-1305)
// and shrinks `limit` -- hence the block size -- in 10 % steps until it succeeds; this library sizes the blocks as if
// the first attempt succeeded and says so here.  A large malloc is an anonymous mmap: the kernel refuses it when it
// exceeds the address-space rlimit, or, by overcommit mode: 0 (heuristic) more than RAM + swap in one piece, 2
// (strict) more than CommitLimit - Committed_AS; mode 1 never refuses.
static bool host_would_refuse(int64_t bytes)
{
	if (bytes <= 0)
		return false;
	struct rlimit rl;
	if (getrlimit(RLIMIT_AS, &rl) == 0 && rl.rlim_cur != RLIM_INFINITY && (uint64_t)bytes > (uint64_t)rl.rlim_cur)
		return true;
	int mode = 0;
	if (FILE *f = fopen("/proc/sys/vm/overcommit_memory", "r")) {
		if (fscanf(f, "%d", &mode) != 1)
			mode = 0;
		fclose(f);
	}
	if (mode == 1)
		return false;
	long long mem_total = 0, swap_total = 0, commit_limit = 0, committed = 0;
	if (FILE *f = fopen("/proc/meminfo", "r")) {
		char key[64];
		long long v;
		while (fscanf(f, "%63s %lld%*[^\n]", key, &v) == 2) {
			if (!strcmp(key, "MemTotal:"))
				mem_total = v;
			else if (!strcmp(key, "SwapTotal:"))
				swap_total = v;
			else if (!strcmp(key, "CommitLimit:"))
				commit_limit = v;
			else if (!strcmp(key, "Committed_AS:"))
				committed = v;
		}
		fclose(f);
	}
	if (!mem_total)
		return false;
	if (mode == 2)
		return bytes > (commit_limit - committed) * 1024;
	return bytes > (mem_total + swap_total) * 1024;
}

extern "C" int lrzgpu_plan(lrzgpu_control *control, int64_t st_size, int64_t *chunk_size)
{
	if (!control || st_size < 0)
		return LRZGPU_E_PARAM;
	Sizing s;
	int r = sizing_for_input(control, st_size, &s);
	if (r)
		return r;
	control->stream_bufsize = s.stream_bufsize;
	control->dictSize_used = s.dict_size;
	control->threads_used = s.threads;
	control->st_size = st_size;
	control->backoff_would_apply = control->malloc_probe ? s.backoff_steps : (host_would_refuse(s.malloc_test) ? 1 : 0);
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
	return abi_guard([&] {
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
	});
}
