// unrzip.cpp -- the read side of the path as an in-library round-trip verifier (SURVEY 8f "next" #1):
// .lrz container walk (reference src/stream.c:1352-1506 open_stream_in, 2023-2195 fill_buffer),
// per-block LZMA decode (lzma_dec.cpp), rzip token replay (src/runzip.c:146-260 unzip_literal /
// unzip_match), chunk CRC-32 and MD5 trailer checks (src/runzip.c:352-440).  Host code: decompression
// is outside the accelerated path; this exists so that a GPU box can check decode(compress(x)) == x
// through the C ABI without the reference.  Subset: what lrzgpu_compress_* writes (lrzip-next 0.14
// magic, no encryption/filters, stored and LZMA blocks, MD5 or no hash).
#include <dlfcn.h>
#include <unistd.h>

#include <atomic>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <new>
#include <thread>
#include <vector>

#include "../../include/lrzgpu.h"
#include "lzma_dec.h"
#include "filters.h"
#include "hashes.h"
#include "md5.h"

using namespace lrzgpu;

namespace {

struct Block {
	int c_type;
	size_t off, c_len, u_len;
};

// CRC-32/IEEE, slice-by-8 (the chunk trailer of src/rzip.c:741-761)
uint32_t crc32_host(uint32_t crc, const uint8_t *p, size_t n)
{
	struct Tables {
		uint32_t t[8][256];
		Tables()
		{
			for (uint32_t i = 0; i < 256; i++) {
				uint32_t r = i;
				for (int j = 0; j < 8; j++)
					r = (r >> 1) ^ (0xEDB88320u & (0u - (r & 1)));
				t[0][i] = r;
			}
			for (uint32_t i = 0; i < 256; i++)
				for (int k = 1; k < 8; k++)
					t[k][i] = (t[k - 1][i] >> 8) ^ t[0][t[k - 1][i] & 0xFF];
		}
	};
	static const Tables tables; // function-local static: initialised once, thread-safe
	const uint32_t(*T)[256] = tables.t;
	crc = ~crc;
	while (n >= 8) {
		uint32_t a, b;
		memcpy(&a, p, 4);
		memcpy(&b, p + 4, 4);
		a ^= crc;
		crc = T[7][a & 0xFF] ^ T[6][(a >> 8) & 0xFF] ^ T[5][(a >> 16) & 0xFF] ^ T[4][a >> 24] ^ T[3][b & 0xFF] ^
		      T[2][(b >> 8) & 0xFF] ^ T[1][(b >> 16) & 0xFF] ^ T[0][b >> 24];
		p += 8;
		n -= 8;
	}
	while (n--)
		crc = (crc >> 8) ^ T[0][(crc ^ *p++) & 0xFF];
	return ~crc;
}

inline uint64_t val(const uint8_t *p, int n)
{
	uint64_t v = 0;
	for (int i = 0; i < n && i < 8; i++)
		v |= (uint64_t)p[i] << (8 * i);
	return v;
}

// zstd blocks (c_type 10, --zstd files): the system libzstd, as src/stream.c:563-590 zstd_decompress_buf
size_t zstd_decompress(void *dst, size_t cap, const void *src, size_t n, bool *ok)
{
	typedef size_t (*Fn)(void *, size_t, const void *, size_t);
	typedef unsigned (*Err)(size_t);
	struct Lib {
		Fn fn = nullptr;
		Err is_err = nullptr;
		Lib()
		{
			void *h = dlopen("libzstd.so.1", RTLD_NOW | RTLD_GLOBAL);
			if (h) {
				fn = (Fn)dlsym(h, "ZSTD_decompress");
				is_err = (Err)dlsym(h, "ZSTD_isError");
			}
		}
	};
	static const Lib lib; // function-local static: bound once, thread-safe
	if (!lib.fn || !lib.is_err) {
		*ok = false;
		return 0;
	}
	const size_t r = lib.fn(dst, cap, src, n);
	*ok = !lib.is_err(r);
	return r;
}

// one stream of a chunk -> its bytes; blocks are decoded by `nthreads` workers
// `limit`: the most bytes this stream can legitimately hold (derived from the file size in the magic);
// every length here comes from the untrusted image, so sums are checked before they can wrap
// `filter` / `delta`: the filter to undo on every block after its back end (literal stream only,
// src/stream.c:1926-1990; stored blocks hold filtered bytes too)
int stream_bytes(const uint8_t *img, const std::vector<Block> &blocks, unsigned lc, unsigned lp, unsigned pb, int nthreads,
		 size_t limit, std::vector<uint8_t> *out, int filter = 0, int delta = 0)
{
	size_t total = 0;
	std::vector<size_t> at(blocks.size());
	for (size_t i = 0; i < blocks.size(); i++) {
		at[i] = total;
		if (blocks[i].u_len > limit - total)
			return LRZGPU_E_FORMAT;
		if (blocks[i].c_type == 3 && blocks[i].c_len != blocks[i].u_len)
			return LRZGPU_E_FORMAT;
		total += blocks[i].u_len;
	}
	out->resize(total);
	std::atomic<size_t> next{0};
	std::atomic<int> err{0};
	auto work = [&] {
		try {
		for (;;) {
			const size_t i = next.fetch_add(1);
			if (i >= blocks.size() || err.load())
				return;
			const Block &b = blocks[i];
			if (b.c_type == 3) {
				if (b.c_len != b.u_len)
					err = LRZGPU_E_FORMAT;
				else
					memcpy(out->data() + at[i], img + b.off, b.c_len);
			} else if (b.c_type == 6) {
				if (lzma_decode_block(img + b.off, b.c_len, out->data() + at[i], b.u_len, lc, lp, pb) != 0)
					err = LRZGPU_E_FORMAT;
			} else if (b.c_type == 10) {
				bool ok = false;
				const size_t r = zstd_decompress(out->data() + at[i], b.u_len, img + b.off, b.c_len, &ok);
				if (!ok || r != b.u_len)
					err = LRZGPU_E_FORMAT;
			} else
				err = LRZGPU_E_PARAM; // other back ends are outside this library
			if (filter && !err.load() && filter_block(filter, delta, out->data() + at[i], b.u_len, false) != 0)
				err = LRZGPU_E_PARAM;
		}
		} catch (const std::bad_alloc &) { // thread body: nothing may escape
			err = LRZGPU_E_NOMEM;
		} catch (...) {
			err = LRZGPU_E_INTERNAL;
		}
	};
	std::vector<std::thread> th;
	const int nt = nthreads < 1 ? 1 : nthreads > (int)blocks.size() ? (int)(blocks.size() ? blocks.size() : 1) : nthreads;
	for (int t = 1; t < nt; t++)
		th.emplace_back(work);
	work();
	for (auto &t : th)
		t.join();
	return err.load();
}

} // namespace

// The checks that read every rebuilt byte once more -- a chunk's CRC-32 and the hash over the whole file
// (src/runzip.c:352-440) -- follow the rebuild on a thread of their own, chunk by chunk in file order, while the next
// chunk's blocks are decoded and replayed: serial MD5 runs at ~1 GB/s, 16 s of a 16 GiB file that used to come after
// everything else.  Only when the output buffer cannot move (an image that states its size); otherwise inline.
struct Checker {
	struct Job {
		const uint8_t *p;
		size_t n;
		uint32_t want_crc;
	};
	std::unique_ptr<Hasher> hasher; // null: chunk CRCs only
	bool threaded = false;
	std::thread th;
	std::mutex mu;
	std::condition_variable cv;
	std::deque<Job> q;
	bool closed = false;
	std::atomic<bool> bad{false};

	void run(const Job &j)
	{
		if (crc32_host(0, j.p, j.n) != j.want_crc)
			bad = true;
		if (hasher)
			hasher->update(j.p, j.n);
	}
	void start()
	{
		threaded = true;
		th = std::thread([this] {
			for (;;) {
				Job j;
				{
					std::unique_lock<std::mutex> lk(mu);
					cv.wait(lk, [this] { return closed || !q.empty(); });
					if (q.empty())
						return;
					j = q.front();
					q.pop_front();
				}
				try {
					run(j);
				} catch (...) {
					bad = true;
				}
			}
		});
	}
	void push(const uint8_t *p, size_t n, uint32_t want_crc)
	{
		const Job j{p, n, want_crc};
		if (!threaded) {
			run(j);
			return;
		}
		std::lock_guard<std::mutex> lk(mu);
		q.push_back(j);
		cv.notify_one();
	}
	void join() // every pushed job done
	{
		if (!threaded || !th.joinable())
			return;
		{
			std::lock_guard<std::mutex> lk(mu);
			closed = true;
		}
		cv.notify_one();
		th.join();
	}
	~Checker() { join(); }
};

static int decompress_impl(const uint8_t *img, int64_t n, uint8_t **out, int64_t *out_len, int host_threads);

extern "C" int lrzgpu_decompress_buffer(const uint8_t *img, int64_t n, uint8_t **out, int64_t *out_len, int host_threads)
{
	try { // the image is untrusted input: absurd sizes in its headers must not escape as C++ exceptions
		return decompress_impl(img, n, out, out_len, host_threads);
	} catch (const std::bad_alloc &) {
		return LRZGPU_E_NOMEM;
	} catch (...) {
		return LRZGPU_E_INTERNAL;
	}
}

static int decompress_impl(const uint8_t *img, int64_t n, uint8_t **out, int64_t *out_len, int host_threads)
{
	if (!img || !out || !out_len || n < 21 + 2)
		return LRZGPU_E_PARAM;
	if (memcmp(img, "LRZI", 4) != 0 || img[4] != 0 || img[5] != 14)
		return LRZGPU_E_FORMAT;
	if (img[15]) // encryption: outside this library
		return LRZGPU_E_PARAM;
	// filter on the literal stream (magic[16], get_filter() of src/lrzip.c:304-338 for 0.13+): bit 7 = delta with
	// its distance 1..16, 32, 48 ... 256 coded as 1..31
	int filter = 0, delta = 0;
	if (img[16] > 128) {
		const int i = img[16] - 128;
		filter = FILTER_DELTA;
		delta = i <= 16 ? i : (i - 15) * 16;
	} else
		filter = img[16];
	if (filter && !filter_supported(filter, delta))
		return LRZGPU_E_PARAM;
	const uint64_t st_size = val(img + 6, 8);
	// the hash after the last chunk: any of the reference's (src/main.c:64-79); 0 = chunk CRCs only
	const int hash_code = img[14];
	const int hash_len = hash_code == 0 ? 0 : hash_length(hash_code);
	if (hash_len < 0)
		return LRZGPU_E_FORMAT;
	const unsigned lc = 3, lp = 0, pb = 2; // LZMA_LC/LP/PB of src/stream.c:450-456
	size_t pos = 21 + img[20];
	struct FreeDeleter {
		void operator()(uint8_t *p) const { free(p); }
	};
	// An image written to STDOUT in several chunks carries no size (the magic went out before the last chunk was
	// known, src/stream.c:1725-1729): the output then grows chunk by chunk by the size each chunk header states
	// (open_stream_in adds them up the same way, src/stream.c:1419)
	const bool sized = st_size != 0;
	size_t dst_cap = sized ? (size_t)st_size : 1;
	std::unique_ptr<uint8_t, FreeDeleter> dst_owner((uint8_t *)malloc(dst_cap));
	uint8_t *dst = dst_owner.get();
	if (!dst)
		return LRZGPU_E_NOMEM;
	if (host_threads <= 0)
		host_threads = (int)std::thread::hardware_concurrency();
	uint64_t at = 0;
	int rc = 0;
	Checker chk; // (declared after the buffer: joined before the buffer goes)
	if (hash_len)
		chk.hasher = make_hasher(hash_code);
	if (sized && st_size >= ((uint64_t)1 << 20) && host_threads > 1)
		chk.start();
	for (;;) {
		if (chk.bad) {
			rc = LRZGPU_E_FORMAT;
			break;
		}
		if (pos + 2 > (size_t)n) {
			rc = LRZGPU_E_FORMAT;
			break;
		}
		const int cb = img[pos], eof = img[pos + 1];
		if (cb < 1 || cb > 8 || pos + 2 + (size_t)cb + 2 * (1 + 3 * (size_t)cb) > (size_t)n) {
			rc = LRZGPU_E_FORMAT;
			break;
		}
		const size_t base = pos + 2 + (size_t)cb, hlen = 1 + 3 * (size_t)cb;
		size_t end = base + 2 * hlen;
		if (!sized) {
			// >= the chunk's bytes: never below one page -- of which a one-byte field keeps only the low byte
			// (src/stream.c:1150-1152, 1747: "Chunk size: smaller than 4,096 bytes" on the read side)
			uint64_t chunk_field = val(img + pos + 2, cb);
			if (chunk_field < 4096)
				chunk_field = 4096;
			if (chunk_field > ((uint64_t)1 << 46) || at + chunk_field < at) {
				rc = LRZGPU_E_FORMAT;
				break;
			}
			if (at + chunk_field > dst_cap) {
				uint8_t *bigger = (uint8_t *)realloc(dst, (size_t)(at + chunk_field));
				if (!bigger) {
					rc = LRZGPU_E_NOMEM;
					break;
				}
				(void)dst_owner.release();
				dst_owner.reset(bigger);
				dst = bigger;
				dst_cap = (size_t)(at + chunk_field);
			}
		}
		const uint64_t out_limit = sized ? st_size : (uint64_t)dst_cap;
		std::vector<Block> blocks[2];
		for (int s = 0; s < 2 && !rc; s++) {
			size_t h = base + (size_t)s * hlen;
			for (;;) {
				if (h + hlen > (size_t)n) {
					rc = LRZGPU_E_FORMAT;
					break;
				}
				Block b;
				b.c_type = img[h];
				b.c_len = (size_t)val(img + h + 1, cb);
				b.u_len = (size_t)val(img + h + 1 + cb, cb);
				const size_t nxt = (size_t)val(img + h + 1 + 2 * cb, cb);
				b.off = h + hlen;
				if (b.c_len) {
					if (b.c_len > (size_t)n - b.off) { // b.off <= n was checked above; no wrap
						rc = LRZGPU_E_FORMAT;
						break;
					}
					blocks[s].push_back(b);
					if (b.off + b.c_len > end)
						end = b.off + b.c_len;
				} else if (h + hlen > end)
					end = h + hlen;
				if (!nxt)
					break;
				if (nxt > (size_t)n - base || base + nxt <= h) { // inside the image, and chains only run forward
					rc = LRZGPU_E_FORMAT;
					break;
				}
				h = base + nxt;
			}
		}
		if (rc)
			break;
		std::vector<uint8_t> s0, s1;
		// what is left of the file bounds both streams: literals byte for byte, tokens by 3 + cb <= 11
		// bytes per token that covers at least one byte, plus the 7-byte tail
		const size_t left = (size_t)(out_limit - at);
		if (left > (size_t)-1 / 16) {
			rc = LRZGPU_E_FORMAT;
			break;
		}
		if ((rc = stream_bytes(img, blocks[0], lc, lp, pb, host_threads, 12 * left + 4096, &s0)) != 0 ||
		    (rc = stream_bytes(img, blocks[1], lc, lp, pb, host_threads, left, &s1, filter, delta)) != 0)
			break;
		// token replay
		size_t i = 0, lit = 0;
		const uint64_t chunk0 = at;
		bool done = false;
		while (!done) {
			if (i + 3 > s0.size()) {
				rc = LRZGPU_E_FORMAT;
				break;
			}
			const int t = s0[i];
			const size_t ln = s0[i + 1] | ((size_t)s0[i + 2] << 8);
			i += 3;
			if (t == 0) {
				if (ln == 0) {
					done = true;
					break;
				}
				if (lit + ln > s1.size() || at + ln > out_limit) {
					rc = LRZGPU_E_FORMAT;
					break;
				}
				memcpy(dst + at, s1.data() + lit, ln);
				lit += ln;
				at += ln;
			} else {
				if (i + (size_t)cb > s0.size()) {
					rc = LRZGPU_E_FORMAT;
					break;
				}
				const uint64_t ofs = val(s0.data() + i, cb);
				i += (size_t)cb;
				if (ofs == 0 || ofs > at - chunk0 || at + ln > out_limit) { // matches stay inside their chunk
					rc = LRZGPU_E_FORMAT;
					break;
				}
				const uint8_t *src = dst + at - ofs;
				if (ofs >= ln)
					memcpy(dst + at, src, ln);
				else
					for (size_t k = 0; k < ln; k++)
						dst[at + k] = src[k];
				at += ln;
			}
		}
		if (rc)
			break;
		if (i + 4 != s0.size() || lit != s1.size()) {
			rc = LRZGPU_E_FORMAT;
			break;
		}
		const uint32_t want = ((uint32_t)s0[i] << 24) | ((uint32_t)s0[i + 1] << 16) | ((uint32_t)s0[i + 2] << 8) | s0[i + 3];
		chk.push(dst + chunk0, (size_t)(at - chunk0), want);
		pos = end;
		if (eof)
			break;
	}
	chk.join();
	if (!rc && chk.bad)
		rc = LRZGPU_E_FORMAT;
	if (!rc && sized && at != st_size)
		rc = LRZGPU_E_FORMAT;
	if (!rc && hash_len) {
		uint8_t dg[64];
		if (pos + (size_t)hash_len != (size_t)n)
			rc = LRZGPU_E_FORMAT;
		else {
			chk.hasher->finish(dg); // (the chunks are the file: every byte went through update() in order)
			if (memcmp(dg, img + pos, (size_t)hash_len) != 0)
				rc = LRZGPU_E_FORMAT;
		}
	} else if (!rc && pos != (size_t)n)
		rc = LRZGPU_E_FORMAT;
	if (rc)
		return rc;
	*out = dst_owner.release();
	*out_len = (int64_t)at;
	return 0;
}

// decompress_file() for the same subset (reference src/lrzip.c decompress_file -> runzip_fd)
extern "C" int lrzgpu_decompress_file(int fd_in, int fd_out, int host_threads)
{
	std::vector<uint8_t> img;
	{
		uint8_t tmp[1 << 16];
		for (;;) {
			const ssize_t r = read(fd_in, tmp, sizeof(tmp));
			if (r < 0)
				return LRZGPU_E_IO;
			if (r == 0)
				break;
			img.insert(img.end(), tmp, tmp + r);
		}
	}
	uint8_t *out = nullptr;
	int64_t n = 0;
	const int rc = lrzgpu_decompress_buffer(img.data(), (int64_t)img.size(), &out, &n, host_threads);
	if (rc)
		return rc;
	int ret = 0;
	for (int64_t o = 0; o < n;) {
		const ssize_t w = write(fd_out, out + o, (size_t)(n - o > (1 << 30) ? (1 << 30) : n - o));
		if (w <= 0) {
			ret = LRZGPU_E_IO;
			break;
		}
		o += w;
	}
	free(out);
	return ret;
}

// get_fileinfo() essentials (reference src/lrzip.c:1069-1460, lrzip-next -i): sizes and block counts
extern "C" int lrzgpu_file_info(const uint8_t *img, int64_t n, lrzgpu_info *info)
{
	if (!img || !info || n < 21 + 2)
		return LRZGPU_E_PARAM;
	if (memcmp(img, "LRZI", 4) != 0)
		return LRZGPU_E_FORMAT;
	memset(info, 0, sizeof(*info));
	info->major = img[4];
	info->minor = img[5];
	info->st_size = (int64_t)val(img + 6, 8);
	info->hash_code = img[14];
	info->lzma = img[17] == 1;
	info->dict_prop = img[18];
	info->rzip_level = img[19] >> 4;
	info->level = img[19] & 15;
	if (img[4] != 0 || img[5] != 14)
		return LRZGPU_E_FORMAT;
	size_t pos = 21 + img[20];
	for (;;) {
		if (pos + 2 > (size_t)n)
			return LRZGPU_E_FORMAT;
		const int cb = img[pos], eof = img[pos + 1];
		if (cb < 1 || cb > 8 || pos + 2 + (size_t)cb + 2 * (1 + 3 * (size_t)cb) > (size_t)n)
			return LRZGPU_E_FORMAT;
		const size_t base = pos + 2 + (size_t)cb, hlen = 1 + 3 * (size_t)cb;
		size_t end = base + 2 * hlen;
		info->chunks++;
		for (int s = 0; s < 2; s++) {
			size_t h = base + (size_t)s * hlen;
			for (;;) {
				if (h + hlen > (size_t)n)
					return LRZGPU_E_FORMAT;
				const int c_type = img[h];
				const size_t c_len = (size_t)val(img + h + 1, cb), u_len = (size_t)val(img + h + 1 + cb, cb);
				const size_t nxt = (size_t)val(img + h + 1 + 2 * cb, cb);
				if (c_len) {
					if (c_len > (size_t)n - (h + hlen))
						return LRZGPU_E_FORMAT;
					info->blocks++;
					if (c_type == 6)
						info->blocks_lzma++;
					info->stream_c_len[s] += (int64_t)c_len;
					info->stream_u_len[s] += (int64_t)u_len;
					if (h + hlen + c_len > end)
						end = h + hlen + c_len;
				}
				if (!nxt)
					break;
				if (nxt > (size_t)n - base || base + nxt <= h)
					return LRZGPU_E_FORMAT;
				h = base + nxt;
			}
		}
		pos = end;
		if (eof)
			break;
	}
	info->compressed_size = n;
	return 0;
}
