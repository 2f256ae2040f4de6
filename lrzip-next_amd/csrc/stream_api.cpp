// stream_api.cpp -- the compress side of the reference's stream layer as a C ABI, call for call
// (src/include/stream.h:14-32; src/stream.c: prepare_streamout_threads 1090-1118, open_stream_out 1140-1348,
// write_stream 2198-2216, flush_buffer 1878-1881 -> clear_buffer 1836-1875 -> compthread 1550-1834,
// close_stream_out 2253-2282, close_streamout_threads 1121-1136) and the per-block back-end dispatch seam
// (`lzma_compress_buf(control, cthread, current_thread)`, src/stream.c:429-494, 1633-1714).
//
// A caller that produces the two rzip streams itself (the reference's own hash_search(), token by token)
// links these instead of stream.c's and gets the same file: blocks are cut at stream_bufsize, handed to a
// ring of workers in flush order, run through the lz4 gate + GPU match finder + host parser, and written
// strictly in hand-off order with the chunk header in front of a chunk's first block and every block header
// patched into its predecessor's next-pointer.  Each worker keeps its device buffers and finder workspace
// for its lifetime; nothing is allocated per block.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <unistd.h>

#include <condition_variable>
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
#include "filters.h"
#include "lzma_mf.h"
#include "pools.h"
#include "stream_layer.h"

using namespace lrzgpu;

namespace {

// what one worker thread owns for as long as it lives
struct BlockBackend {
	int device = 0;
	hipStream_t s = nullptr;
	DevBuf d_block, d_small; // the block's bytes; gate descriptor + result
	MfWorkspace *ws = nullptr;
	double ws_per_pos = 0;
	size_t ws_n = 0;
	~BlockBackend() { close(); }
	void close()
	{
		WorkspacePool::get().give_mf(ws, ws_per_pos, device);
		ws = nullptr;
		StreamPool::get().give(s); // parked, never destroyed (pools.h)
		s = nullptr;
		d_block.release();
		d_small.release();
	}
	int open(int dev)
	{
		device = dev;
		if (select_device(dev))
			return LRZGPU_E_NODEVICE;
		if (!s && !(s = pooled_stream(dev)))
			return LRZGPU_E_HIP;
		return 0;
	}
	int upload(const uint8_t *p, size_t n)
	{
		if (d_block.cap < n + 256 && !d_block.alloc(n + 256, device))
			return LRZGPU_E_NOMEM;
		if (!d_small.p && !d_small.alloc(256, device))
			return LRZGPU_E_NOMEM;
		if ((n && hipMemcpyAsync(d_block.p, p, n, hipMemcpyHostToDevice, s) != hipSuccess) ||
		    hipMemsetAsync(d_block.p + n, 0, 256, s) != hipSuccess || stream_wait(s) != hipSuccess)
			return LRZGPU_E_HIP;
		return 0;
	}
	// LZ4_compress_default size of the first in_len bytes of the uploaded block
	int lz4_size(int in_len, int d_len)
	{
		Lz4Job job{d_block.p, in_len, d_len, 0};
		Lz4Job *dj = (Lz4Job *)d_small.p;
		int *dr = (int *)(d_small.p + 64), res = -1;
		if (hipMemcpyAsync(dj, &job, sizeof(job), hipMemcpyHostToDevice, s) != hipSuccess || lz4_sizes_device(dj, 1, dr, s) != 0 ||
		    hipMemcpyAsync(&res, dr, sizeof(int), hipMemcpyDeviceToHost, s) != hipSuccess || stream_wait(s) != hipSuccess)
			return LRZGPU_E_HIP;
		return res;
	}
	// lz4_compresses(): 0 = leave the block alone, 1..100 = percentage, < 0 error
	int gate(int64_t n, int threshold)
	{
		int err = 0;
		int v = lz4_compresses_decision(n, threshold, [&](int in_len, int d_len) {
			int r = lz4_size(in_len, d_len);
			if (r < 0) {
				err = r;
				return 0;
			}
			return r;
		});
		return err ? err : v;
	}
	// LzmaCompress() of the uploaded block (host copy `src`): SRes
	int lzma(const LzmaParams &lp, const uint8_t *src, size_t n, uint8_t *dest, size_t cap, size_t *out_len)
	{
		const bool pack = lp.dict_size <= (1u << 25) && lp.fb <= 65;
		double per_pos = 16;
		unsigned long long total = 0;
		for (int attempt = 0;; attempt++) {
			if (!ws || ws_n < n) {
				WorkspacePool::get().give_mf(ws, ws_per_pos, device);
				ws = WorkspacePool::get().take_mf(n ? n : 1, per_pos, device, &ws_per_pos);
				if (!ws)
					return LZ_ERROR_MEM;
				ws_n = ws->max_n;
			}
			int r = mf_run_device(ws, d_block.p, n, lp.dict_size, (uint32_t)lp.fb, lp.cut(), s, &total, pack ? 2 : 1, lp.fast);
			if (r == 0)
				break;
			if (r == -4 && attempt < 3) {
				per_pos = ws_per_pos * 3;
				mf_workspace_destroy(ws);
				ws = nullptr;
				continue;
			}
			return r == -4 ? LZ_ERROR_MEM : LZ_ERROR_PARAM;
		}
		const size_t words = pack ? (size_t)(total / 2) : (size_t)total;
		RawBuf<uint8_t> counts;
		RawBuf<uint32_t> pairs;
		counts.alloc(n ? n : 1);
		pairs.alloc(words ? words : 1);
		if ((n && hipMemcpyAsync(counts.data(), ws->counts, n, hipMemcpyDeviceToHost, s) != hipSuccess) ||
		    (words && hipMemcpyAsync(pairs.data(), ws->pool_out, words * 4, hipMemcpyDeviceToHost, s) != hipSuccess) ||
		    stream_wait(s) != hipSuccess)
			return LZ_ERROR_MEM;
		MatchLists ml;
		ml.counts = counts.data();
		ml.pairs = pairs.data();
		ml.packed = pack;
		ml.tail_flags = true;
		return lzma_encode_block(lp, src, n, ml, dest, cap, out_len);
	}
};

struct ZstdFn {
	size_t (*compress)(void *, size_t, const void *, size_t, int) = nullptr;
	unsigned (*is_error)(size_t) = nullptr;
	ZstdFn()
	{
		void *h = dlopen("libzstd.so.1", RTLD_NOW | RTLD_GLOBAL);
		if (h) {
			compress = (size_t(*)(void *, size_t, const void *, size_t, int))dlsym(h, "ZSTD_compress");
			is_error = (unsigned (*)(size_t))dlsym(h, "ZSTD_isError");
		}
	}
};

// One block through the back end the control asks for.  In: s_buf/s_len (malloc'd, owned by *t).  Out, as the
// reference's <x>_compress_buf leaves a compress_thread: compressed -> s_buf replaced, c_len/c_type set;
// left alone -> untouched (c_type stays CTYPE_NONE).  0, or -1 when a resource was missing (the caller may
// wait for its predecessor and try once more, src/stream.c:1667-1714).
int backend_block(lrzgpu_control *control, const Sizing &sz, BlockBackend &be, lrzgpu_compress_thread *t)
{
	if (sz.no_compress || t->s_len < 64)
		return 0; // src/stream.c:1633
	int rc = be.open(control->device);
	if (rc)
		return -1;
	if (!sz.zstd || sz.lz4_test) {
		if (be.upload(t->s_buf, (size_t)t->s_len))
			return -1;
	}
	if (sz.lz4_test) {
		const int pct = be.gate(t->s_len, sz.threshold);
		if (pct < 0)
			return -1;
		if (pct == 0)
			return 0;
	}
	if (sz.zstd) { // zstd_compress_buf(), src/stream.c:167-230
		static const ZstdFn z;
		if (!z.compress || !z.is_error)
			return -1;
		const size_t cap = ((size_t)t->s_len + kPage - 1) / kPage * kPage;
		uint8_t *c_buf = (uint8_t *)malloc(cap);
		if (!c_buf)
			return -1;
		const size_t r = z.compress(c_buf, cap, t->s_buf, (size_t)t->s_len, sz.zstd_level);
		if (z.is_error(r) || (int64_t)r >= t->c_len) {
			const bool fatal = z.is_error(r) && (size_t)0 - r != 70; // anything but "does not fit"
			free(c_buf);
			return fatal ? -1 : 0;
		}
		free(t->s_buf);
		t->s_buf = c_buf;
		t->c_len = (int64_t)r;
		t->c_type = CTYPE_ZSTD;
		return 0;
	}
	for (;;) { // lzma_compress_buf(), src/stream.c:429-494
		LzmaParams lp;
		static std::mutex level_mu; // several workers share one control: the level is read and lowered under a lock
		int level;
		{
			std::lock_guard<std::mutex> lk(level_mu);
			level = control->compression_level;
		}
		if (lzma_normalize(lp, level, control->dictSize_used ? control->dictSize_used : sz.dict_size, 3, 0, 2, level < 7 ? 32 : 64) != LZ_OK)
			return -1;
		size_t dlen = (size_t)((double)t->s_len * 1.02);
		dlen = (dlen + kPage - 1) / kPage * kPage;
		uint8_t *c_buf = (uint8_t *)malloc(dlen);
		if (!c_buf)
			return -1;
		size_t out_len = 0;
		const int r = be.lzma(lp, t->s_buf, (size_t)t->s_len, c_buf, dlen, &out_len);
		if (r != LZ_OK) {
			free(c_buf);
			if (r == LZ_ERROR_MEM && level > 1) {
				// "Can't allocate enough RAM for compression window, trying smaller": one step per failed level,
				// however many workers fail at it together
				std::lock_guard<std::mutex> lk(level_mu);
				if (control->compression_level == level)
					control->compression_level = level - 1;
				continue;
			}
			return r == LZ_ERROR_OUTPUT_EOF ? 0 : -1;
		}
		lzma_write_props(lp, control->lzma_properties);
		if ((int64_t)out_len >= t->c_len) {
			free(c_buf);
			return 0; // incompressible: stays CTYPE_NONE
		}
		free(t->s_buf);
		t->s_buf = c_buf;
		t->c_len = (int64_t)out_len;
		t->c_type = CTYPE_LZMA;
		return 0;
	}
}

// ---- the ring (the statics of src/stream.c:87-91: cthreads[], output_thread, stream_bufsize ...) --------------
struct StreamOut;

struct Task {
	lrzgpu_compress_thread ct{};
	StreamOut *sinfo = nullptr;
	uint64_t ticket = 0;
};
static_assert(sizeof(void *) == 8, "64-bit host");

struct StreamOut { // struct stream_info, src/include/lrzip_private.h:592-620 (compress side)
	lrzgpu_control *control = nullptr;
	int fd = -1, num_streams = 2, chunk_bytes = 0, eof = 0;
	int64_t size = 0, bufsize = 0;
	// stream buffers: malloc()ed like the reference's (src/stream.c:1337-1345), handed to the worker as they are
	// (clear_buffer 1836-1875: "the stream buffer has been given to the thread, allocate a new one") and freed by it
	uint8_t *buf[2] = {nullptr, nullptr};
	int64_t buflen[2] = {0, 0};
	~StreamOut()
	{
		free(buf[0]);
		free(buf[1]);
	}
	// writer state, touched only by the thread whose turn it is
	bool header_written = false;
	int64_t initial_pos = 0, cur_pos = 0, last_head[2] = {0, 0};
	int pending = 0; // blocks handed off and not yet written
	bool closed = false;
};

struct Ring {
	std::mutex mu;
	std::condition_variable cv_work, cv_turn, cv_room;
	std::deque<Task *> queue;
	std::vector<std::thread> workers;
	std::vector<std::unique_ptr<StreamOut>> sinfos;
	uint64_t next_ticket = 0, output_ticket = 0;
	int in_flight = 0, slots = 1;
	bool closing = false, prepared = false;
	int err = 0;
	Sizing sz;
	bool sized = false;
	int filter_flag = 0, filter_delta = 0; // latched from the control with the sizing
	int fd = -1;
	int64_t file_pos = -1; // next byte the ordered writer puts out
	lrzgpu_control *control = nullptr;
	static Ring &get()
	{
		static Ring r;
		return r;
	}
};

int pwrite_all(int fd, const uint8_t *p, size_t n, int64_t off)
{
	while (n) {
		ssize_t w = pwrite(fd, p, n > ((size_t)1 << 30) ? ((size_t)1 << 30) : n, (off_t)off);
		if (w <= 0)
			return -1;
		p += w;
		n -= (size_t)w;
		off += w;
	}
	return 0;
}
void le_val(uint8_t *p, int64_t v, int n)
{
	for (int i = 0; i < n; i++)
		p[i] = i < 8 ? (uint8_t)((uint64_t)v >> (8 * i)) : 0;
}

// the write half of compthread(), src/stream.c:1716-1821; called in ticket order
int write_block(Ring &R, Task *t)
{
	StreamOut *s = t->sinfo;
	const int cb = s->chunk_bytes;
	uint8_t h[64];
	if (!s->header_written) {
		s->header_written = true;
		h[0] = (uint8_t)cb;
		h[1] = (uint8_t)s->eof;
		le_val(h + 2, s->size, cb);
		if (pwrite_all(R.fd, h, (size_t)(2 + cb), R.file_pos))
			return LRZGPU_E_IO;
		R.file_pos += 2 + cb;
		s->initial_pos = R.file_pos;
		s->cur_pos = 0;
		for (int j = 0; j < s->num_streams; j++) {
			s->last_head[j] = s->cur_pos + 1 + cb * 2;
			memset(h, 0, sizeof(h));
			h[0] = CTYPE_NONE;
			if (pwrite_all(R.fd, h, (size_t)(1 + 3 * cb), s->initial_pos + s->cur_pos))
				return LRZGPU_E_IO;
			s->cur_pos += 1 + cb * 3;
		}
	}
	const int sn = t->ct.streamno;
	le_val(h, s->cur_pos, cb);
	if (pwrite_all(R.fd, h, (size_t)cb, s->initial_pos + s->last_head[sn]))
		return LRZGPU_E_IO;
	s->last_head[sn] = s->cur_pos + 1 + cb * 2;
	h[0] = t->ct.c_type;
	le_val(h + 1, t->ct.c_len, cb);
	le_val(h + 1 + cb, t->ct.s_len, cb);
	le_val(h + 1 + 2 * cb, 0, cb);
	if (pwrite_all(R.fd, h, (size_t)(1 + 3 * cb), s->initial_pos + s->cur_pos))
		return LRZGPU_E_IO;
	s->cur_pos += 1 + cb * 3;
	if (t->ct.c_len && pwrite_all(R.fd, t->ct.s_buf, (size_t)t->ct.c_len, s->initial_pos + s->cur_pos))
		return LRZGPU_E_IO;
	s->cur_pos += t->ct.c_len;
	R.file_pos = s->initial_pos + s->cur_pos;
	return 0;
}

void drop_sinfo(Ring &R, StreamOut *so) // R.mu held
{
	for (size_t i = 0; i < R.sinfos.size(); i++)
		if (R.sinfos[i].get() == so) {
			R.sinfos[i] = std::move(R.sinfos.back());
			R.sinfos.pop_back();
			return;
		}
}

void worker_main(Ring *Rp)
{
	Ring &R = *Rp;
	BlockBackend be;
	for (;;) {
		Task *t = nullptr;
		bool failed_already;
		{
			std::unique_lock<std::mutex> lk(R.mu);
			R.cv_work.wait(lk, [&] { return !R.queue.empty() || R.closing; });
			if (R.queue.empty())
				return;
			t = R.queue.front();
			R.queue.pop_front();
			failed_already = R.err != 0;
		}
		int rc = 0;
		try {
			// compthread, src/stream.c:1587-1628: the filter runs over a literal block before whatever back end
			// follows (stored blocks included); host converters here -- the whole-file driver filters in HBM
			if (!failed_already && R.filter_flag && t->ct.streamno == 1 && t->ct.s_len &&
			    filter_block(R.filter_flag, R.filter_delta, t->ct.s_buf, (size_t)t->ct.s_len, true) != 0)
				rc = LRZGPU_E_PARAM;
			int r = (failed_already || rc) ? 0 : backend_block(R.control, R.sz, be, &t->ct);
			if (r) { // "Unable to compress in parallel, waiting for previous thread to complete before trying again"
				std::unique_lock<std::mutex> lk(R.mu);
				R.cv_turn.wait(lk, [&] { return R.output_ticket == t->ticket; });
				lk.unlock();
				if (backend_block(R.control, R.sz, be, &t->ct))
					rc = LRZGPU_E_NOMEM; // the reference is fatal() here
			}
		} catch (...) {
			rc = LRZGPU_E_NOMEM;
		}
		std::unique_lock<std::mutex> lk(R.mu);
		R.cv_turn.wait(lk, [&] { return R.output_ticket == t->ticket; });
		if (!rc && !R.err)
			rc = write_block(R, t);
		if (rc && !R.err)
			R.err = rc;
		free(t->ct.s_buf);
		t->sinfo->pending--;
		delete t;
		R.output_ticket++;
		R.in_flight--;
		R.cv_turn.notify_all();
		R.cv_room.notify_all();
	}
}

int hand_off(Ring &R, StreamOut *s, int streamno) // clear_buffer(), src/stream.c:1836-1875
{
	std::unique_ptr<Task> t(new Task());
	t->sinfo = s;
	t->ct.streamno = streamno;
	t->ct.s_len = t->ct.c_len = s->buflen[streamno];
	t->ct.c_type = CTYPE_NONE;
	// the buffer itself goes to the worker (no copy); the stream gets a new one with its next byte
	t->ct.s_buf = s->buf[streamno] ? s->buf[streamno] : (uint8_t *)malloc(1);
	if (!t->ct.s_buf)
		return LRZGPU_E_NOMEM;
	s->buf[streamno] = nullptr;
	s->buflen[streamno] = 0;
	std::unique_lock<std::mutex> lk(R.mu);
	R.cv_room.wait(lk, [&] { return R.in_flight < R.slots || R.err; }); // the slot's semaphore
	if (R.err) {
		free(t->ct.s_buf);
		return R.err;
	}
	t->ticket = R.next_ticket++;
	R.in_flight++;
	s->pending++;
	R.queue.push_back(t.release());
	R.cv_work.notify_one();
	return 0;
}

} // namespace

extern "C" int lrzgpu_lzma_compress_buf(lrzgpu_control *control, lrzgpu_compress_thread *cthread, int current_thread)
{
	(void)current_thread;
	if (!control || !cthread || !cthread->s_buf || cthread->s_len < 0)
		return -1;
	try {
		Sizing sz;
		if (compute_sizing(control, control->st_size, &sz))
			return -1;
		thread_local BlockBackend be; // the calling compthread's own device buffers, kept between blocks
		return backend_block(control, sz, be, cthread);
	} catch (...) {
		return -1;
	}
}

extern "C" int lrzgpu_prepare_streamout_threads(lrzgpu_control *control)
{
	if (!control || control->threads < 1)
		return LRZGPU_E_PARAM;
	Ring &R = Ring::get();
	std::lock_guard<std::mutex> lk(R.mu);
	if (R.prepared)
		return LRZGPU_E_PARAM;
	R.control = control;
	R.closing = false;
	R.err = 0;
	R.next_ticket = R.output_ticket = 0;
	R.in_flight = 0;
	R.sized = false;
	R.fd = -1;
	R.file_pos = -1;
	// threads + 1 slots so that one can be filled while the others compress; one when nothing is compressed
	R.slots = control->threads > 1 ? control->threads + 1 : control->threads;
	if (control->flags & LRZGPU_FLAG_NO_COMPRESS)
		R.slots = 1;
	int nw = control->host_threads > 0 ? control->host_threads : R.slots;
	const int hw = (int)std::thread::hardware_concurrency();
	if (hw > 0 && nw > hw)
		nw = hw;
	if (nw > R.slots)
		nw = R.slots;
	try {
		for (int i = 0; i < nw; i++)
			R.workers.emplace_back(worker_main, &R);
	} catch (...) {
		return LRZGPU_E_NOMEM;
	}
	R.prepared = true;
	return 0;
}

extern "C" void *lrzgpu_open_stream_out(lrzgpu_control *control, int f, unsigned int n, int64_t chunk_limit, char cbytes)
{
	Ring &R = Ring::get();
	if (!control || n != 2 || cbytes < 1 || cbytes > 8 || !R.prepared)
		return nullptr;
	try {
		std::unique_ptr<StreamOut> s(new StreamOut());
		s->control = control;
		s->fd = f;
		s->num_streams = (int)n;
		s->chunk_bytes = cbytes;
		s->size = chunk_limit < kPage ? kPage : chunk_limit;
		s->eof = control->eof ? 1 : 0;
		std::lock_guard<std::mutex> lk(R.mu);
		if (!R.sized) { // "This block only has to be done once since memory never changes"
			if (compute_sizing(control, control->st_size, &R.sz))
				return nullptr;
			R.sized = true;
			control->stream_bufsize = R.sz.stream_bufsize;
			control->dictSize_used = R.sz.dict_size;
			control->threads_used = R.sz.threads;
			R.fd = f;
			R.file_pos = (int64_t)lseek(f, 0, SEEK_CUR);
			if (R.file_pos < 0)
				return nullptr; // the ordered writer patches headers in place: the output must be seekable
		} else if (f != R.fd) {
			return nullptr; // one output file per prepare/close cycle: the ordered writer owns its position
		}
		s->bufsize = R.sz.stream_bufsize;
		StreamOut *raw = s.get();
		R.sinfos.push_back(std::move(s));
		return raw;
	} catch (...) {
		return nullptr;
	}
}

extern "C" int lrzgpu_flush_buffer(lrzgpu_control *control, void *ss, int stream)
{
	(void)control;
	StreamOut *s = (StreamOut *)ss;
	if (!s || stream < 0 || stream >= s->num_streams || s->closed)
		return LRZGPU_E_PARAM;
	try {
		return hand_off(Ring::get(), s, stream);
	} catch (...) {
		return LRZGPU_E_NOMEM;
	}
}

extern "C" int lrzgpu_write_stream(lrzgpu_control *control, void *ss, int streamno, const uint8_t *p, int64_t len)
{
	StreamOut *s = (StreamOut *)ss;
	if (!s || streamno < 0 || streamno >= s->num_streams || len < 0 || s->closed)
		return LRZGPU_E_PARAM;
	try {
		while (len) { // src/stream.c:2198-2216
			if (!s->buf[streamno] && !(s->buf[streamno] = (uint8_t *)malloc((size_t)s->bufsize)))
				return LRZGPU_E_NOMEM;
			int64_t k = s->bufsize - s->buflen[streamno];
			if (k > len)
				k = len;
			memcpy(s->buf[streamno] + s->buflen[streamno], p, (size_t)k);
			s->buflen[streamno] += k;
			p += k;
			len -= k;
			if (s->buflen[streamno] == s->bufsize) {
				int r = lrzgpu_flush_buffer(control, ss, streamno);
				if (r)
					return r;
			}
		}
		return 0;
	} catch (...) {
		return LRZGPU_E_NOMEM;
	}
}

extern "C" int lrzgpu_close_stream_out(lrzgpu_control *control, void *ss)
{
	StreamOut *s = (StreamOut *)ss;
	if (!s || s->closed)
		return LRZGPU_E_PARAM;
	int rc = 0;
	for (int i = 0; i < s->num_streams && !rc; i++) // unconditional, zero-length blocks included (src/stream.c:2258-2259)
		rc = lrzgpu_flush_buffer(control, ss, i);
	// both buffers are with the workers now; the handle goes when its last block has been written
	Ring &R = Ring::get();
	std::lock_guard<std::mutex> lk(R.mu);
	s->closed = true;
	if (s->pending == 0)
		drop_sinfo(R, s);
	return rc;
}

extern "C" int lrzgpu_close_streamout_threads(lrzgpu_control *control)
{
	(void)control;
	Ring &R = Ring::get();
	std::vector<std::thread> workers;
	int err;
	{
		std::unique_lock<std::mutex> lk(R.mu);
		if (!R.prepared)
			return LRZGPU_E_PARAM;
		R.cv_room.wait(lk, [&] { return R.in_flight == 0; }); // every slot's semaphore, src/stream.c:1121-1136
		R.closing = true;
		R.cv_work.notify_all();
		workers.swap(R.workers);
		err = R.err;
	}
	for (auto &w : workers)
		w.join();
	std::lock_guard<std::mutex> lk(R.mu);
	if (R.fd >= 0 && R.file_pos >= 0 && !err && lseek(R.fd, (off_t)R.file_pos, SEEK_SET) < 0)
		err = LRZGPU_E_IO; // the caller appends the hash and rewrites the magic from here
	R.sinfos.clear();
	R.prepared = false;
	return err;
}
