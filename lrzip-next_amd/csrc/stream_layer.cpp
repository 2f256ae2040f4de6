// stream_layer.cpp -- see stream_layer.h.  Host-only logic, no device code.
#include <cstdlib>
#include "stream_layer.h"

#include <atomic>
#include <cstring>
#include <thread>
#include <utility>

namespace lrzgpu {

namespace {
constexpr int64_t ONE_MB = 1048576;
constexpr int64_t STREAM_BUFSIZE = 10 * ONE_MB; // src/include/lrzip_private.h:16
constexpr int64_t CHUNK_MULTIPLE = 100 * ONE_MB; // src/rzip.c:48

int64_t round_to_page(int64_t v) // src/util.c:190-195
{
	v -= v % kPage;
	return v ? v : kPage;
}
int64_t round_up_page(int64_t v) // src/util.c:197-204
{
	int64_t r = v % kPage;
	return r ? v + kPage - r : v;
}
uint32_t lzma2_dic_from_prop(unsigned p) // src/include/lrzip_private.h:236
{
	return p == 40 ? 0xFFFFFFFFu : ((uint32_t)(2 | (p & 1)) << (p / 2 + 11));
}
unsigned lzma2_prop_from_dic(uint32_t d)
{
	unsigned i = 0;
	for (; i <= 40; i++)
		if (d <= lzma2_dic_from_prop(i))
			break;
	return i;
}
uint32_t level_dict(int level) // src/util.c:108-127
{
	switch (level) {
	case 1: case 2: case 3: return 1u << (level * 2 + 16);
	case 4: case 5: case 6: return 1u << (level + 19);
	case 7: return 1u << 25;
	case 8: return 1u << 26;
	case 9: return 1u << 27;
	default: return 1u << 24;
	}
}
int64_t overhead_for(uint32_t dict) { return ((int64_t)dict * 23 / 2) + 6 * ONE_MB + 16384; } // src/util.c:131
} // namespace

int compute_sizing(const lrzgpu_control *c, int64_t st_size, Sizing *out, int64_t chunk_limit_arg)
{
	Sizing s;
	s.level = c->compression_level;
	s.rzip_level = c->rzip_compression_level ? c->rzip_compression_level : c->compression_level; // src/main.c:779-780
	s.no_compress = (c->flags & LRZGPU_FLAG_NO_COMPRESS) != 0;
	s.lz4_test = (c->flags & LRZGPU_FLAG_THRESHOLD) != 0 && !s.no_compress && !c->filter_flag; // src/main.c:858-861
	s.nobemt = (c->flags & LRZGPU_FLAG_NOBEMT) != 0;
	s.threshold = c->threshold;
	if (s.level < 1 || s.level > 9 || s.rzip_level < 1 || s.rzip_level > 9 || c->threads < 1 || c->ramsize <= 0)
		return LRZGPU_E_PARAM;
	s.zstd = (c->flags & LRZGPU_FLAG_ZSTD) != 0 && !s.no_compress;
	if (s.zstd) {
		// zstd level <-> strategy <-> rzip level, src/main.c:87 (zstd_compression_level[]), 692-711, 822-828
		static const int by_level[10] = {-1, 2, 4, 5, 7, 12, 15, 17, 18, 22};
		if (c->zstd_level) {
			if (c->zstd_level < 1 || c->zstd_level > 22)
				return LRZGPU_E_PARAM;
			s.zstd_level = c->zstd_level;
			for (int st = 1; st <= 9; st++)
				if (s.zstd_level <= by_level[st]) {
					s.zstd_strategy = st;
					if (!c->rzip_compression_level)
						s.rzip_level = st; // --zstd-level sets the rzip level to the strategy
					break;
				}
		} else {
			s.zstd_level = by_level[s.level];
			s.zstd_strategy = s.level;
		}
	}
	const bool lzma = !s.no_compress && !s.zstd;
	s.dict_size = c->dictSize ? c->dictSize : level_dict(s.level);
	s.overhead = lzma ? overhead_for(s.dict_size) : 0;

	// setup_ram(), src/util.c:179-188: a sixth when the compressed file is held back for stdout
	const int64_t usable_ram = c->stdout_mode ? c->ramsize / 6 : c->ramsize / 3;
	const int64_t maxram = round_to_page(usable_ram);

	// rzip_fd(), src/rzip.c:999-1013
	int64_t max_mmap = round_to_page(maxram);
	int64_t max_chunk = c->window ? c->window * CHUNK_MULTIPLE : c->ramsize / 3 * 2;
	if (max_mmap > max_chunk)
		max_mmap = max_chunk;
	// (STDIN: control->st_size is still 0 at this point of rzip_fd(), so the window is never rounded)
	if (!c->stdin_mode && max_chunk < st_size)
		max_chunk = round_to_page(max_chunk);
	s.max_chunk = max_chunk;
	s.max_mmap = max_mmap;

	// prepare_streamout_threads(), src/stream.c:1099-1102
	s.threads = c->threads;
	if (s.threads > 1)
		s.threads++;
	if (s.no_compress)
		s.threads = 1;

	// open_stream_out() first call, src/stream.c:1169-1331
	int64_t chunk_limit = chunk_limit_arg >= 0 ? chunk_limit_arg : (max_chunk < st_size ? max_chunk : st_size);
	if (chunk_limit < kPage)
		chunk_limit = kPage;
	const int testbufs = s.no_compress ? 1 : 2;
	int64_t limit = usable_ram / testbufs;
	if (lzma) {
		const int save_threads = s.threads;
		int thread_limit = s.threads >= c->processors / 2 ? s.threads / 2 : s.threads;
		unsigned exponent = lzma2_prop_from_dic(s.dict_size);
		const uint32_t save_dict = s.dict_size;
		const unsigned save_exp = exponent;
		bool found = false;
		for (;;) {
			do {
				for (s.threads = save_threads; s.threads >= thread_limit; s.threads--)
					if (limit >= s.overhead * s.threads / testbufs) {
						found = true;
						break;
					}
				if (found)
					break;
				exponent -= 1;
				s.dict_size = lzma2_dic_from_prop(exponent);
				s.overhead = overhead_for(s.dict_size);
			} while (s.dict_size > (1u << 24));
			if (!found && thread_limit > 1) {
				thread_limit--;
				s.dict_size = save_dict;
				exponent = save_exp;
				s.overhead = overhead_for(s.dict_size);
				continue;
			}
			break;
		}
	}
	if (s.threads < 1)
		return LRZGPU_E_PARAM; // ramsize too small for any LZMA thread (the reference divides by zero here)
	if (st_size > 0 && st_size < limit)
		limit = st_size > STREAM_BUFSIZE ? st_size : STREAM_BUFSIZE;
	else if (limit > chunk_limit)
		limit = chunk_limit;
	// retest_malloc, src/stream.c:1290-1305: the reference asks the host for limit + overhead * threads bytes and takes a
	// tenth off `limit` for as long as malloc() refuses -- so its block size depends on what the process may allocate
	// (address-space rlimit, overcommit policy).  With control->malloc_probe the same probe is made here, the same
	// steps, the same "cannot even get 100 MB" failure; without it the first probe is taken to succeed (sizes are then a
	// function of the parameters alone; lrzgpu_plan() says when this host would have refused).  (The pointer is
	// volatile: a malloc / free pair with no use in between is something compilers delete.)
	s.malloc_test = limit + s.overhead * s.threads;
	while (c->malloc_probe) {
		void *volatile probe = malloc((size_t)(limit + s.overhead * s.threads));
		if (probe) {
			free(probe);
			break;
		}
		limit = limit / 10 * 9;
		s.backoff_steps++;
		if (limit < 100000000)
			return LRZGPU_E_NOMEM;
	}
	if (lzma && limit / s.threads > STREAM_BUFSIZE) {
		int64_t a = s.overhead - (int64_t)s.dict_size;
		s.stream_bufsize = round_up_page((limit > a ? limit : a) / s.threads);
	} else {
		int64_t a = limit / s.threads;
		if (a < STREAM_BUFSIZE)
			a = STREAM_BUFSIZE;
		s.stream_bufsize = round_up_page(limit < a ? limit : a);
	}
	*out = s;
	return 0;
}

int sizing_for_input(const lrzgpu_control *c, int64_t n, Sizing *out)
{
	if (!c->stdin_mode)
		return compute_sizing(c, n, out);
	Sizing pre;
	int r = compute_sizing(c, 0, &pre);
	if (r)
		return r;
	const int64_t first = pre.max_mmap < n ? pre.max_mmap : n;
	r = compute_sizing(c, first, out, first);
	if (!r)
		out->max_chunk = pre.max_mmap;
	return r;
}

void chunk_sizes_for(const lrzgpu_control *c, const Sizing &s, int64_t n, std::vector<int64_t> *sizes)
{
	sizes->clear();
	int64_t len = n;
	bool stdin_eof = false;
	while (sizes->empty() || len > 0 || (c->stdin_mode && !stdin_eof)) {
		const int64_t size = s.max_chunk < len ? s.max_chunk : len;
		if (size < s.max_chunk)
			stdin_eof = true; // a short read: control->eof = st->stdin_eof = 1
		sizes->push_back(size);
		len -= size;
	}
}

void block_order(const std::vector<uint8_t> &stream0, int chunk_bytes, int64_t stream1_len, int64_t bufsize,
		 std::vector<BlockRef> *blocks)
{
	int64_t fill[2] = {0, 0}, start[2] = {0, 0};
	auto put = [&](int s, int64_t n) {
		while (n) {
			int64_t k = bufsize - fill[s];
			if (k > n)
				k = n;
			fill[s] += k;
			n -= k;
			if (fill[s] == bufsize) {
				blocks->push_back(BlockRef{s, start[s], bufsize});
				start[s] += bufsize;
				fill[s] = 0;
			}
		}
	};
	size_t i = 0;
	const size_t n0 = stream0.size();
	while (i + 3 <= n0) {
		const int head = stream0[i];
		const int64_t len = stream0[i + 1] | ((int64_t)stream0[i + 2] << 8);
		put(0, 3);
		i += 3;
		if (head == 1) {
			put(0, chunk_bytes);
			i += (size_t)chunk_bytes;
		} else {
			if (len == 0) { // terminator: the 4 CRC bytes follow
				put(0, (int64_t)(n0 - i));
				i = n0;
				break;
			}
			put(1, len);
		}
	}
	(void)stream1_len;
	// close_stream_out(): stream 0 then stream 1, unconditionally (zero-length blocks included)
	blocks->push_back(BlockRef{0, start[0], fill[0]});
	blocks->push_back(BlockRef{1, start[1], fill[1]});
}

static void put_val(uint8_t *o, size_t at, int64_t v, int n)
{
	for (int i = 0; i < n; i++)
		o[at + i] = i < 8 ? (uint8_t)((uint64_t)v >> (8 * i)) : 0;
}

size_t chunk_image_size(int cb, const std::vector<DoneBlock> &blocks)
{
	size_t total = 2 + (size_t)cb + 2 * (1 + 3 * (size_t)cb);
	for (const DoneBlock &b : blocks)
		total += 1 + 3 * (size_t)cb + b.payload.size();
	return total;
}

void write_chunk_raw(uint8_t *o, int cb, bool eof, int64_t chunk_size, const std::vector<DoneBlock> &blocks)
{
	o[0] = (uint8_t)cb;
	o[1] = eof ? 1 : 0;
	put_val(o, 2, chunk_size < kPage ? kPage : chunk_size, cb); // sinfo->size, src/stream.c:1150-1152, 1747
	uint8_t *base = o + 2 + cb; // initial_pos: every offset in the headers is relative to it
	int64_t cur_pos = 0, last_head[2];
	for (int j = 0; j < 2; j++) { // src/stream.c:1755-1769
		last_head[j] = cur_pos + 1 + cb * 2;
		base[cur_pos] = CTYPE_NONE;
		put_val(base, (size_t)cur_pos + 1, 0, cb * 3);
		cur_pos += 1 + cb * 3;
	}
	std::vector<std::pair<size_t, const DoneBlock *>> big; // payloads worth a thread of their own
	size_t big_bytes = 0;
	for (const DoneBlock &b : blocks) { // src/stream.c:1772-1821
		put_val(base, (size_t)last_head[b.streamno], cur_pos, cb);
		last_head[b.streamno] = cur_pos + 1 + cb * 2;
		base[cur_pos] = (uint8_t)b.c_type;
		put_val(base, (size_t)cur_pos + 1, (int64_t)b.payload.size(), cb);
		put_val(base, (size_t)cur_pos + 1 + cb, b.s_len, cb);
		put_val(base, (size_t)cur_pos + 1 + 2 * cb, 0, cb);
		cur_pos += 1 + cb * 3;
		if (b.payload.size() >= ((size_t)8 << 20)) {
			big.push_back(std::make_pair((size_t)cur_pos, &b));
			big_bytes += b.payload.size();
		} else if (!b.payload.empty())
			memcpy(base + cur_pos, b.payload.data(), b.payload.size());
		cur_pos += (int64_t)b.payload.size();
	}
	// a chunk of stored blocks is gigabytes of payload: one core moves ~4 GB/s, a few threads the rest of what the
	// memory system gives (the committer is alone at this point of a chunk: the last chunk's layout is a serial tail)
	const unsigned nt = big_bytes >= ((size_t)256 << 20) ? 8u : 1u;
	std::atomic<size_t> next{0};
	auto work = [&] {
		for (;;) {
			const size_t k = next.fetch_add(1);
			if (k >= big.size())
				return;
			memcpy(base + big[k].first, big[k].second->payload.data(), big[k].second->payload.size());
		}
	};
	std::vector<std::thread> th;
	try {
		for (unsigned t = 1; t < nt; t++)
			th.emplace_back(work);
	} catch (...) { // no thread to be had: the caller's does it all
	}
	work();
	for (auto &t : th)
		t.join();
}

void write_chunk(std::vector<uint8_t> *outp, int cb, bool eof, int64_t chunk_size, const std::vector<DoneBlock> &blocks)
{
	const size_t at = outp->size();
	outp->resize(at + chunk_image_size(cb, blocks));
	write_chunk_raw(outp->data() + at, cb, eof, chunk_size, blocks);
}

void write_magic(uint8_t magic[21], const Sizing &s, int64_t st_size)
{
	memset(magic, 0, 21);
	memcpy(magic, "LRZI", 4);
	magic[4] = 0;  // LRZIP_MAJOR_VERSION
	magic[5] = 14; // LRZIP_MINOR_VERSION
	for (int i = 0; i < 8; i++)
		magic[6 + i] = (uint8_t)((uint64_t)st_size >> (8 * i));
	magic[14] = 1; // hash_code: MD5
	if (s.zstd) { // src/lrzip.c:177-183
		magic[17] = (uint8_t)((s.zstd_strategy << 4) + 4);
		magic[18] = (uint8_t)s.zstd_level;
	} else if (!s.no_compress) {
		magic[17] = 1;
		magic[18] = (uint8_t)lzma2_prop_from_dic(s.dict_size);
	}
	magic[19] = (uint8_t)((s.rzip_level << 4) + s.level);
	magic[20] = 0; // comment length
}

// magic[16], write_magic() src/lrzip.c:146-156: the filter's code, the delta distance folded into 128 + 1..31
uint8_t filter_magic_byte(int filter_flag, int delta)
{
	if (filter_flag == 128)
		return (uint8_t)(128 + (delta <= 16 ? delta : delta / 16 + 15));
	return (uint8_t)filter_flag;
}

void write_magic_for(uint8_t magic[21], const lrzgpu_control *c, const Sizing &s, int64_t st_size, size_t n_chunks)
{
	// "else if (control->eof)" (src/lrzip.c:141-144): to a file the magic is written after the last chunk; to stdout
	// with the first block of the first chunk (src/stream.c:1725-1729), when eof is set only if that chunk is the last
	write_magic(magic, s, (c->stdout_mode && n_chunks > 1) ? 0 : st_size);
	magic[14] = (uint8_t)c->hash_code;
	magic[16] = filter_magic_byte(c->filter_flag, c->delta);
}

} // namespace lrzgpu
