// stream_in.cpp -- the read side of the reference's stream layer as a C ABI, call for call (src/include/stream.h:
// open_stream_in 26, read_stream 29, close_stream_in 31, write_1g / read_1g 20-21, put_fdout 32; src/stream.c:
// open_stream_in 1352-1506, fill_buffer 2023-2195, ucompthread 1883-2021, read_stream 2220-2250, close_stream_in
// 2299-2319).  For a caller that replays the rzip tokens itself -- the reference's runzip_chunk(), src/runzip.c:
// 139-370: it reads the chunk_bytes byte, opens the two streams of the chunk, pulls token headers and match offsets
// from stream 0 and literal bytes from stream 1, and closes the chunk, which leaves the fd at the next one.
//
// Host code: decompression is the verifier side of this library, not an accelerated path.  Like the
// reference, blocks are fetched and decoded ahead of the reader by worker threads (one for the token stream, up to
// control->threads for the literal stream); the container walk is the format's: a chunk's block headers form one
// singly linked list per stream, offsets relative to the position after the chunk header.
// Subset: what this library writes and reads -- stored, LZMA and zstd blocks, no encryption.
#include <dlfcn.h>
#include <unistd.h>

#include <cerrno>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <future>
#include <memory>
#include <new>
#include <vector>

#include "../../include/lrzgpu.h"
#include "filters.h"
#include "lzma_dec.h"

using namespace lrzgpu;

namespace {

int read_all(int fd, uint8_t *p, size_t n)
{
	while (n) {
		const ssize_t r = read(fd, p, n > ((size_t)1 << 30) ? ((size_t)1 << 30) : n); // one_g pieces, src/stream.c:832
		if (r < 0 && errno == EINTR)
			continue;
		if (r <= 0)
			return -1;
		p += r;
		n -= (size_t)r;
	}
	return 0;
}
int64_t le_val(const uint8_t *p, int n)
{
	uint64_t v = 0;
	for (int i = 0; i < n && i < 8; i++)
		v |= (uint64_t)p[i] << (8 * i);
	return (int64_t)v;
}

size_t zstd_decompress(void *dst, size_t cap, const void *src, size_t n, bool *ok) // src/stream.c:563-590
{
	typedef size_t (*Fn)(void *, size_t, const void *, size_t);
	typedef unsigned (*Err)(size_t);
	struct Lib {
		Fn fn = nullptr;
		Err is_err = nullptr;
		Lib()
		{
			if (void *h = dlopen("libzstd.so.1", RTLD_NOW | RTLD_GLOBAL)) {
				fn = (Fn)dlsym(h, "ZSTD_decompress");
				is_err = (Err)dlsym(h, "ZSTD_isError");
			}
		}
	};
	static const Lib lib;
	if (!lib.fn || !lib.is_err) {
		*ok = false;
		return 0;
	}
	const size_t r = lib.fn(dst, cap, src, n);
	*ok = !lib.is_err(r);
	return r;
}

struct FreeDeleter {
	void operator()(uint8_t *p) const { free(p); }
};
typedef std::unique_ptr<uint8_t, FreeDeleter> Bytes;

// one block: what ucompthread() does with it (back end, then the filter of magic[16] on literal blocks)
struct Decoded {
	Bytes buf;
	int64_t len = 0;
	int err = 0;
};
Decoded decode_block(int c_type, Bytes src, int64_t c_len, int64_t u_len, int streamno, int filter, int delta)
{
	Decoded d;
	try {
		if (c_type == 3) { // CTYPE_NONE: the buffer as it is
			if (c_len != u_len)
				d.err = LRZGPU_E_FORMAT;
			d.buf = std::move(src);
		} else {
			d.buf.reset((uint8_t *)malloc(u_len ? (size_t)u_len : 1));
			if (!d.buf)
				d.err = LRZGPU_E_NOMEM;
			else if (c_type == 6) {
				if (lzma_decode_block(src.get(), (size_t)c_len, d.buf.get(), (size_t)u_len, 3, 0, 2) != 0)
					d.err = LRZGPU_E_FORMAT;
			} else if (c_type == 10) {
				bool ok = false;
				const size_t r = zstd_decompress(d.buf.get(), (size_t)u_len, src.get(), (size_t)c_len, &ok);
				if (!ok || r != (size_t)u_len)
					d.err = LRZGPU_E_FORMAT;
			} else
				d.err = LRZGPU_E_PARAM; // bzip2 / gzip / lzo / zpaq / bzip3 back ends are outside this library
		}
		d.len = u_len;
		if (!d.err && filter && streamno == 1 && u_len && filter_block(filter, delta, d.buf.get(), (size_t)u_len, false) != 0)
			d.err = LRZGPU_E_PARAM; // src/stream.c:1926-1990
	} catch (...) {
		d.err = LRZGPU_E_INTERNAL;
	}
	return d;
}

struct StreamIn { // struct stream_info + struct stream, read side (src/include/lrzip_private.h:592-620)
	int fd = -1, num_streams = 2, chunk_bytes = 0;
	int64_t size = 0, initial_pos = 0, total_read = 0;
	int filter = 0, delta = 0;
	int ahead_limit[2] = {1, 1}; // s[0].total_threads = 1, s[1].total_threads = total_threads - 1 (src/stream.c:1394-1395)
	int64_t ram_alloced = 0, maxram = 0;
	struct S {
		int64_t last_head = 0;
		bool eos = false;
		Bytes buf;
		int64_t buflen = 0, bufp = 0;
		std::deque<std::future<Decoded>> ahead;
		std::deque<int64_t> ahead_len;
	} s[2];
};

// fill_buffer(), src/stream.c:2023-2195: fetch the next block(s) of the stream and start decoding them, as many ahead as
// the stream has workers and memory allows; then take the oldest one
int fill_buffer(StreamIn *si, int streamno)
{
	StreamIn::S &s = si->s[streamno];
	s.buf.reset();
	s.buflen = s.bufp = 0;
	const int cb = si->chunk_bytes;
	const int64_t hlen = 1 + 3 * (int64_t)cb;
	while (!s.eos && (int)s.ahead.size() < si->ahead_limit[streamno] && (s.ahead.empty() || si->ram_alloced < si->maxram)) {
		uint8_t h[32];
		if (lseek(si->fd, (off_t)(si->initial_pos + s.last_head), SEEK_SET) < 0 || read_all(si->fd, h, (size_t)hlen))
			return -1;
		si->total_read += hlen;
		const int c_type = h[0];
		const int64_t c_len = le_val(h + 1, cb), u_len = le_val(h + 1 + cb, cb), last_head = le_val(h + 1 + 2 * cb, cb);
		if (c_len == 0 && u_len == 0 && streamno == 1 && last_head == 0) { // "an empty match block at the end"
			s.eos = true;
			break;
		}
		// zero-length blocks exist in what the compress side writes (close_stream_out flushes both streams
		// unconditionally): they carry no bytes and only move the chain on
		if (c_len < 0 || u_len < 0 || last_head < 0 || (last_head && last_head <= s.last_head))
			return -1;
		if (c_len == 0) {
			if (u_len != 0)
				return -1;
			s.last_head = last_head;
			if (!last_head)
				s.eos = true;
			continue;
		}
		Bytes src((uint8_t *)malloc((size_t)(c_len > u_len ? c_len : u_len)));
		if (!src || read_all(si->fd, src.get(), (size_t)c_len))
			return -1;
		si->total_read += c_len;
		si->ram_alloced += u_len;
		s.last_head = last_head;
		if (!last_head)
			s.eos = true;
		const int filter = si->filter, delta = si->delta;
		uint8_t *raw = src.release();
		try {
			s.ahead.push_back(std::async(std::launch::async, [=] { return decode_block(c_type, Bytes(raw), c_len, u_len, streamno, filter, delta); }));
		} catch (...) {
			free(raw);
			return -1;
		}
		s.ahead_len.push_back(u_len);
	}
	if (s.ahead.empty())
		return 0; // end of the stream: buflen stays 0
	Decoded d = s.ahead.front().get();
	s.ahead.pop_front();
	si->ram_alloced -= s.ahead_len.front();
	s.ahead_len.pop_front();
	if (d.err)
		return -1;
	s.buf = std::move(d.buf);
	s.buflen = d.len;
	s.bufp = 0;
	return 0;
}

} // namespace

extern "C" void *lrzgpu_open_stream_in(lrzgpu_control *control, int f, int n, char chunk_bytes)
{
	if (!control || n != 2 || chunk_bytes < 1 || chunk_bytes > 8)
		return nullptr;
	try {
		std::unique_ptr<StreamIn> si(new StreamIn());
		si->fd = f;
		si->num_streams = n;
		si->chunk_bytes = chunk_bytes;
		if (control->filter_flag) {
			if (!filter_supported(control->filter_flag, control->delta))
				return nullptr;
			si->filter = control->filter_flag;
			si->delta = control->filter_flag == FILTER_DELTA ? control->delta : 0;
		}
		// "one thread dedicated to stream 0, and one more thread than CPUs to keep them busy"
		const int total_threads = control->threads > 1 ? control->threads + 2 : control->threads + 1;
		si->ahead_limit[0] = 1;
		si->ahead_limit[1] = total_threads - 1 < 1 ? 1 : total_threads - 1;
		si->maxram = control->ramsize > 0 ? control->ramsize / 3 : (int64_t)1 << 30;
		const int cb = chunk_bytes;
		uint8_t h[64];
		// the eof flag and the chunk size follow the chunk_bytes byte the caller has read (src/stream.c:1398-1424)
		if (read_all(f, h, (size_t)(1 + cb)))
			return nullptr;
		control->eof = h[0];
		si->size = le_val(h + 1, cb);
		if (si->size < 0)
			return nullptr;
		control->st_size += si->size;
		const off_t here = lseek(f, 0, SEEK_CUR);
		if (here < 0)
			return nullptr;
		si->initial_pos = (int64_t)here;
		const int64_t hlen = 1 + 3 * (int64_t)cb;
		for (int i = 0; i < n; i++) { // the initial header of every stream: type NONE, no bytes, the first real header's offset
			if (read_all(f, h, (size_t)hlen))
				return nullptr;
			si->total_read += hlen;
			const int64_t v1 = le_val(h + 1, cb), v2 = le_val(h + 1 + cb, cb);
			si->s[i].last_head = le_val(h + 1 + 2 * cb, cb);
			if (h[0] != 3 || v1 || v2 || si->s[i].last_head < 0)
				return nullptr; // "Unexpected initial tag / c_len / u_len in streams"
			if (!si->s[i].last_head)
				si->s[i].eos = true;
		}
		return si.release();
	} catch (...) {
		return nullptr;
	}
}

// "read some data from a stream. Return number of bytes read, or -1 on failure" (src/stream.c:2218-2250)
extern "C" int64_t lrzgpu_read_stream(lrzgpu_control *control, void *ss, int streamno, uint8_t *p, int64_t len)
{
	(void)control;
	StreamIn *si = (StreamIn *)ss;
	if (!si || streamno < 0 || streamno >= si->num_streams || len < 0 || (len && !p))
		return -1;
	try {
		StreamIn::S &s = si->s[streamno];
		int64_t ret = 0;
		while (len) {
			int64_t k = s.buflen - s.bufp;
			if (k > len)
				k = len;
			if (k > 0) {
				memcpy(p, s.buf.get() + s.bufp, (size_t)k);
				s.bufp += k;
				p += k;
				len -= k;
				ret += k;
			}
			if (len && s.bufp == s.buflen) {
				if (fill_buffer(si, streamno))
					return -1;
				if (s.bufp == s.buflen)
					break;
			}
		}
		return ret;
	} catch (...) {
		return -1;
	}
}

// leaves the fd after the chunk's last block: where the next chunk (or the hash) starts (src/stream.c:2299-2319)
extern "C" int lrzgpu_close_stream_in(lrzgpu_control *control, void *ss)
{
	(void)control;
	StreamIn *si = (StreamIn *)ss;
	if (!si)
		return -1;
	int rc = 0;
	try {
		for (auto &s : si->s)
			for (auto &f : s.ahead)
				(void)f.get();
		if (lseek(si->fd, (off_t)(si->initial_pos + si->total_read), SEEK_SET) < 0)
			rc = -1;
	} catch (...) {
		rc = -1;
	}
	delete si;
	return rc;
}

// ssize_t write_1g(control, buf, len): everything to control->fd_out in pieces of at most 1 GiB (src/stream.c:817-850)
extern "C" int64_t lrzgpu_write_1g(lrzgpu_control *control, const void *buf, int64_t len)
{
	if (!control || len < 0 || (len && !buf))
		return -1;
	const uint8_t *p = (const uint8_t *)buf;
	int64_t total = 0;
	while (len > 0) {
		const ssize_t w = write(control->fd_out, p, (size_t)(len > ((int64_t)1 << 30) ? ((int64_t)1 << 30) : len));
		if (w < 0 && errno == EINTR)
			continue;
		if (w <= 0)
			return -1;
		p += w;
		len -= w;
		total += w;
	}
	return total;
}

// ssize_t read_1g(control, fd, buf, len): as much as there is, up to len (src/stream.c:897-945)
extern "C" int64_t lrzgpu_read_1g(lrzgpu_control *control, int fd, void *buf, int64_t len)
{
	(void)control;
	if (len < 0 || (len && !buf))
		return -1;
	uint8_t *p = (uint8_t *)buf;
	int64_t total = 0;
	while (len > 0) {
		const ssize_t r = read(fd, p, (size_t)(len > ((int64_t)1 << 30) ? ((int64_t)1 << 30) : len));
		if (r < 0 && errno == EINTR)
			continue;
		if (r < 0)
			return -1;
		if (r == 0)
			return total;
		p += r;
		len -= r;
		total += r;
	}
	return total;
}

// ssize_t put_fdout(control, offset_buf, ret): one piece to the output (src/stream.c:802-815; the reference's temporary
// output buffer for STDOUT is the caller's business here)
extern "C" int64_t lrzgpu_put_fdout(lrzgpu_control *control, const void *offset_buf, int64_t ret)
{
	return lrzgpu_write_1g(control, offset_buf, ret);
}

// i64 get_readseek(control, fd): where the read side stands in its input (src/stream.c:1078-1088; runzip_chunk prints
// it before every chunk header, src/runzip.c:293).  The reference's other branch -- STDIN spooled into a temporary
// input buffer, control->in_ofs -- is its command-line front end's; a caller of this library reads from an fd.
extern "C" int64_t lrzgpu_get_readseek(lrzgpu_control *control, int fd)
{
	(void)control;
	const off_t at = lseek(fd, 0, SEEK_CUR);
	return at == (off_t)-1 ? -1 : (int64_t)at;
}
