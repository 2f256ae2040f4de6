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
//
// Where it lives (round 5: one 2 600-line file until then):
//   driver.cpp         this file: the C ABI of the whole-file path (compress buffer / file / chunks, assemble, plan,
//                      container store), sinks, control defaults
//   scan_run.cpp       struct Run: readers, hash thread, scanners (scan_chunk: the scan of a chunk and what it hands
//                      to the back end while it runs), the committer
//   pipeline.h         Job / ChunkCtx / Pipeline (queues, routing, early-start bookkeeping) / Feeder (gate batches)
//   gpu_worker.cpp     the GPU worker thread: finder runs on whole blocks and on the prefixes of early blocks
//   encoder_worker.cpp the encoder thread: parser + range coder (or zstd) per block, staged lists of early blocks
//   pipeline.cpp       trace clock, CPU seconds by thread role
#include "pipeline.h"

using namespace lrzgpu;

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
	// (the walk's son slots are 32-bit `2 * node + side` words: 2^31 positions -- mf_run_device has the same guard; a
	// part with 320 GB or more would otherwise advertise blocks the finder cannot hold: ADVICE r5)
	const double cap = 2147483632.0; // 0x7FFFFFF0
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

