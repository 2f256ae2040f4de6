/* container_oracle.c -- ORACLE (test infrastructure; see lrzo.h).
 *
 * CPU restatement of the lrzip-next 0.14 compress driver and container layout:
 *   rzip_fd chunk loop / sizing         reference src/rzip.c:922-1264 (999-1020, 1041-1186)
 *   setup_overhead / setup_ram          reference src/util.c:103-188
 *   open_stream_out sizing              reference src/stream.c:1140-1348
 *   write_stream / flush order          reference src/stream.c:2198-2216, 1836-1881, 2253-2282
 *   compthread chunk/block headers      reference src/stream.c:1550-1834
 *   lzma_compress_buf                   reference src/stream.c:429-494
 *   write_magic                         reference src/lrzip.c:131-208
 * The reference orders output with a ring of compthreads; here blocks are queued in
 * flush order, compressed by a worker pool and written in order (same bytes).
 */
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "lrzo.h"

#define ONE_MB 1048576LL
#define STREAM_BUFSIZE (ONE_MB * 10)
#define CHUNK_MULTIPLE (100 * ONE_MB)
#define PAGE 4096LL
#define CTYPE_NONE 3
#define CTYPE_LZMA 6
#define SZ_ERROR_OUTPUT_EOF 7

void lrzo_params_default(lrzo_params *p)
{
	memset(p, 0, sizeof(*p));
	p->compression_level = 7;
	p->threads = 1;
	p->processors = 1;
	p->ramsize = 80 * 100 * ONE_MB;
	p->lz4_test = 1;
	p->threshold = 100;
	p->workers = 1;
}

static i64 round_to_page(i64 v)
{
	v -= v % PAGE;
	return v ? v : PAGE;
}
static i64 round_up_page(i64 v)
{
	i64 rem = v % PAGE;
	return rem ? v + PAGE - rem : v;
}
/* reference src/include/lrzip_private.h:236-245 */
static uint32_t lzma2_dic_from_prop(unsigned p) { return p == 40 ? 0xFFFFFFFFu : ((uint32_t)(2 | (p & 1)) << (p / 2 + 11)); }
static unsigned lzma2_prop_from_dic(uint32_t d)
{
	unsigned i;
	for (i = 0; i <= 40; i++)
		if (d <= lzma2_dic_from_prop(i))
			break;
	return i;
}
static uint32_t dict_for_level(int L) /* src/util.c:108-127 */
{
	if (L >= 1 && L <= 3) return 1u << (L * 2 + 16);
	if (L >= 4 && L <= 6) return 1u << (L + 19);
	if (L == 7) return 1u << 25;
	if (L == 8) return 1u << 26;
	if (L == 9) return 1u << 27;
	return 1u << 24;
}
static i64 lzma_overhead(uint32_t dict) { return ((i64)dict * 23 / 2) + (6 * ONE_MB) + 16384; }

/* ---- block queue ---------------------------------------------------- */
struct block {
	uchar *buf;
	i64 s_len, c_len;
	int streamno, c_type, chunk_no, done;
};

struct outbuf { uchar *p; i64 len, cap; };
static void ob_reserve(struct outbuf *o, i64 need)
{
	if (need <= o->cap)
		return;
	while (o->cap < need)
		o->cap = o->cap ? o->cap * 2 : (1 << 20);
	o->p = realloc(o->p, (size_t)o->cap);
	if (!o->p)
		abort();
}
static void ob_write_at(struct outbuf *o, i64 pos, const void *src, i64 n)
{
	ob_reserve(o, pos + n);
	memcpy(o->p + pos, src, (size_t)n);
	if (pos + n > o->len)
		o->len = pos + n;
}
static void ob_val_at(struct outbuf *o, i64 pos, i64 v, int n)
{
	uchar b[8];
	int i;
	for (i = 0; i < 8; i++)
		b[i] = (uchar)((uint64_t)v >> (8 * i));
	ob_write_at(o, pos, b, n);
}

struct driver {
	const lrzo_params *prm;
	lrzo_lzma_fn lzma;
	int level, threads, lz4_test, zstd_level;
	uint32_t dict_size;
	i64 bufsize;

	/* per-chunk stream state */
	const uchar *chunk;
	uchar *sbuf[2];
	i64 sblen[2];
	int chunk_no;

	/* queue */
	pthread_mutex_t mu;
	pthread_cond_t cv;
	struct block *blocks;
	i64 nblocks, capblocks, next_job;
	int closing;
	lrzo_file_stats *fs;
};

static void (*g_filter)(unsigned char *, size_t);
void lrzo_set_filter(void (*convert)(unsigned char *, size_t)) { g_filter = convert; }
static size_t (*g_zstd_compress)(void *, size_t, const void *, size_t, int);
void lrzo_set_zstd(size_t (*compress)(void *, size_t, const void *, size_t, int)) { g_zstd_compress = compress; }

static void compress_block(struct driver *d, struct block *b)
{
	b->c_type = CTYPE_NONE;
	b->c_len = b->s_len;
	/* compthread, src/stream.c:1587-1628: filters run over stream 1 whatever the back end (even none) */
	if (d->prm->filter_flag && b->streamno == 1 && g_filter)
		g_filter(b->buf, (size_t)b->s_len);
	if (d->prm->no_compress || b->c_len < 64)
		return;
	if (d->lz4_test && !lrzo_lz4_compresses(b->buf, b->s_len, d->prm->threshold))
		return;
	if (d->prm->zstd) { /* zstd_compress_buf, src/stream.c:167-230 */
		size_t dlen = (size_t)round_up_page(b->s_len), r;
		uchar *c_buf = malloc(dlen);
		if (!c_buf || !g_zstd_compress)
			abort();
		r = g_zstd_compress(c_buf, dlen, b->buf, (size_t)b->s_len, d->zstd_level);
		if (r > (size_t)-200) { /* ZSTD_isError */
			if ((size_t)0 - r != 70) { /* only dstSize_tooSmall means "incompressible" */
				fprintf(stderr, "oracle: ZSTD_compress failed %zu\n", (size_t)0 - r);
				abort();
			}
			free(c_buf);
			return;
		}
		if ((i64)r >= b->c_len) {
			free(c_buf);
			return;
		}
		free(b->buf);
		b->buf = c_buf;
		b->c_len = (i64)r;
		b->c_type = 10; /* CTYPE_ZSTD */
		return;
	}
	{
		size_t dlen = (size_t)round_up_page((i64)(size_t)(b->s_len * 1.02));
		size_t prop_size = 5;
		uchar props[8];
		uchar *c_buf = malloc(dlen);
		int ret;
		if (!c_buf)
			abort();
		ret = d->lzma(c_buf, &dlen, b->buf, (size_t)b->s_len, props, &prop_size, d->level, d->dict_size,
			      3, 0, 2, d->level < 7 ? 32 : 64, (d->threads > 1 && d->prm->nobemt) ? 1 : 2);
		if (ret == SZ_ERROR_OUTPUT_EOF || (ret == 0 && (i64)dlen >= b->c_len)) {
			free(c_buf);
			return;
		}
		if (ret != 0) {
			fprintf(stderr, "oracle: LzmaCompress failed %d\n", ret);
			abort();
		}
		free(b->buf);
		b->buf = c_buf;
		b->c_len = (i64)dlen;
		b->c_type = CTYPE_LZMA;
	}
}

static void *worker(void *arg)
{
	struct driver *d = arg;
	for (;;) {
		struct block tmp;
		i64 j;
		pthread_mutex_lock(&d->mu);
		while (d->next_job >= d->nblocks && !d->closing)
			pthread_cond_wait(&d->cv, &d->mu);
		if (d->next_job >= d->nblocks) {
			pthread_mutex_unlock(&d->mu);
			return NULL;
		}
		j = d->next_job++;
		tmp = d->blocks[j]; /* blocks[] may be realloc'd by the producer: work on a copy */
		pthread_mutex_unlock(&d->mu);
		compress_block(d, &tmp);
		tmp.done = 1;
		pthread_mutex_lock(&d->mu);
		d->blocks[j] = tmp;
		pthread_mutex_unlock(&d->mu);
	}
}

static void queue_block(struct driver *d, int streamno, int newbuf)
{
	struct block b;
	memset(&b, 0, sizeof(b));
	b.buf = d->sbuf[streamno];
	b.s_len = d->sblen[streamno];
	b.streamno = streamno;
	b.chunk_no = d->chunk_no;
	pthread_mutex_lock(&d->mu);
	if (d->nblocks == d->capblocks) {
		d->capblocks = d->capblocks ? d->capblocks * 2 : 64;
		d->blocks = realloc(d->blocks, sizeof(struct block) * (size_t)d->capblocks);
		if (!d->blocks)
			abort();
	}
	d->blocks[d->nblocks++] = b;
	pthread_cond_signal(&d->cv);
	pthread_mutex_unlock(&d->mu);
	if (newbuf) {
		d->sbuf[streamno] = malloc((size_t)d->bufsize);
		if (!d->sbuf[streamno])
			abort();
		d->sblen[streamno] = 0;
	}
}

static void sink_put0(void *ctx, const uchar *p, i64 len)
{
	struct driver *d = ctx;
	while (len) {
		i64 n = d->bufsize - d->sblen[0];
		if (n > len)
			n = len;
		memcpy(d->sbuf[0] + d->sblen[0], p, (size_t)n);
		d->sblen[0] += n;
		p += n;
		len -= n;
		if (d->sblen[0] == d->bufsize)
			queue_block(d, 0, 1);
	}
}
static void sink_put1(void *ctx, i64 off, i64 len)
{
	struct driver *d = ctx;
	while (len) {
		i64 n = d->bufsize - d->sblen[1];
		if (n > len)
			n = len;
		memcpy(d->sbuf[1] + d->sblen[1], d->chunk + off, (size_t)n);
		d->sblen[1] += n;
		off += n;
		len -= n;
		if (d->sblen[1] == d->bufsize)
			queue_block(d, 1, 1);
	}
}

struct plan_out { int threads; uint32_t dict_size; i64 bufsize, max_chunk, max_mmap; };

/* setup_overhead / setup_ram / rzip_fd sizing / prepare_streamout_threads / open_stream_out */
static int make_plan(const lrzo_params *prm, i64 n, struct plan_out *po)
{
	struct { int threads, level; uint32_t dict_size; i64 bufsize; } d;
	i64 maxram, usable_ram, max_mmap, max_chunk, overhead, limit;
	const int lzma_on = !prm->no_compress && !prm->zstd;
	d.level = prm->compression_level;
	/* setup_overhead / setup_ram */
	d.dict_size = prm->dict_size ? prm->dict_size : dict_for_level(d.level);
	overhead = lzma_on ? lzma_overhead(d.dict_size) : 0;
	usable_ram = prm->stdout_mode ? prm->ramsize / 6 : prm->ramsize / 3; /* src/util.c:179-188 */
	maxram = round_to_page(usable_ram);

	/* rzip_fd sizing, src/rzip.c:999-1020.  STDIN: control->st_size is still 0 here, so max_chunk is never
	 * rounded, and every pass takes chunk_size = mmap_size = max_mmap (1075); `n` is then the FIRST chunk's
	 * size: what control->st_size holds when open_stream_out() sizes the blocks (mmap_stdin 835) */
	max_mmap = round_to_page(maxram);
	if (prm->window)
		max_chunk = prm->window * CHUNK_MULTIPLE;
	else
		max_chunk = prm->ramsize / 3 * 2;
	if (max_mmap > max_chunk)
		max_mmap = max_chunk;
	if (!prm->stdin_mode && max_chunk < n)
		max_chunk = round_to_page(max_chunk);
	po->max_mmap = max_mmap;

	/* prepare_streamout_threads, src/stream.c:1099-1102 */
	d.threads = prm->threads;
	if (d.threads > 1)
		d.threads++;
	if (prm->no_compress)
		d.threads = 1;

	/* open_stream_out first-call sizing, src/stream.c:1169-1331 (chunk_limit = first chunk size) */
	{
		i64 first_chunk = max_chunk < n ? max_chunk : n;
		i64 chunk_limit = first_chunk < PAGE ? PAGE : first_chunk;
		int testbufs = prm->no_compress ? 1 : 2;
		int save_threads = d.threads;
		limit = usable_ram / testbufs;
		if (lzma_on) {
			int thread_limit = d.threads >= prm->processors / 2 ? d.threads / 2 : d.threads;
			unsigned exponent = lzma2_prop_from_dic(d.dict_size);
			uint32_t save_dict = d.dict_size;
			unsigned save_exp = exponent;
			int set = 0;
		retry:
			do {
				for (d.threads = save_threads; d.threads >= thread_limit; d.threads--)
					if (limit >= overhead * d.threads / testbufs) {
						set = 1;
						break;
					}
				if (set)
					break;
				exponent -= 1;
				d.dict_size = lzma2_dic_from_prop(exponent);
				overhead = lzma_overhead(d.dict_size);
			} while (d.dict_size > (1u << 24));
			if (!set && thread_limit > 1) {
				thread_limit--;
				d.dict_size = save_dict;
				exponent = save_exp;
				overhead = lzma_overhead(d.dict_size);
				goto retry;
			}
		}
		if (d.threads < 1)
			return -1; /* -m too small for one LZMA thread: the reference divides by zero below (src/stream.c:1316) */
		if (n > 0 && n < limit)
			limit = n > STREAM_BUFSIZE ? n : STREAM_BUFSIZE;
		else if (limit > chunk_limit)
			limit = chunk_limit;
		/* retest_malloc, src/stream.c:1290-1305: a tenth off `limit` for as long as the host refuses the allocation
		 * (only when asked for: by default block sizes must not depend on the machine the tests run on) */
		while (prm->malloc_probe) {
			void *volatile probe = malloc((size_t)(limit + overhead * d.threads));
			if (probe) {
				free(probe);
				break;
			}
			limit = limit / 10 * 9;
			if (limit < 100000000)
				return -2;
		}
		if (lzma_on && limit / d.threads > STREAM_BUFSIZE) {
			i64 a = overhead - (i64)d.dict_size;
			d.bufsize = round_up_page((limit > a ? limit : a) / d.threads);
		} else {
			i64 a = limit / d.threads;
			if (a < STREAM_BUFSIZE)
				a = STREAM_BUFSIZE;
			d.bufsize = round_up_page(limit < a ? limit : a);
		}
	}
	po->threads = d.threads;
	po->dict_size = d.dict_size;
	po->bufsize = d.bufsize;
	po->max_chunk = max_chunk;
	return 0;
}

int lrzo_plan(const lrzo_params *prm, i64 n, lrzo_file_stats *fs)
{
	struct plan_out po;
	if (prm->stdin_mode) { /* blocks are sized from the first chunk, and every chunk is max_mmap bytes */
		if (make_plan(prm, 0, &po))
			return -1;
		n = po.max_mmap < n ? po.max_mmap : n;
	}
	if (make_plan(prm, n, &po))
		return -1;
	memset(fs, 0, sizeof(*fs));
	fs->stream_bufsize = po.bufsize;
	fs->threads_used = po.threads;
	fs->dict_size = po.dict_size;
	return 0;
}

int lrzo_compress_buffer(const lrzo_params *prm, const uchar *in, i64 n, lrzo_lzma_fn lzma,
			 uchar **out, i64 *out_len, lrzo_file_stats *fs)
{
	struct driver d;
	struct outbuf ob = {0};
	lrzo_file_stats lfs;
	lrzo_md5 md5;
	uint64_t hx[256];
	i64 victim_round = 0;
	int rzip_level, lzma_on, nworkers, w, zstd_strategy = 0;
	pthread_t *tids;
	i64 max_chunk, len, *chunk_sizes = NULL, nchunks = 0;
	int *chunk_cbytes = NULL;

	memset(&d, 0, sizeof(d));
	memset(&lfs, 0, sizeof(lfs));
	d.prm = prm;
	d.lzma = lzma;
	d.fs = &lfs;
	d.level = prm->compression_level;
	rzip_level = prm->rzip_level ? prm->rzip_level : prm->compression_level;
	lzma_on = !prm->no_compress && !prm->zstd;
	d.lz4_test = prm->lz4_test && !prm->no_compress && !prm->filter_flag; /* src/main.c:858-861 */
	if (lzma_on && !lzma)
		return -1;
	if (prm->zstd && !prm->no_compress) {
		/* zstd level <-> strategy <-> rzip level: src/main.c:87, 692-711, 822-828 */
		static const int by_level[10] = {-1, 2, 4, 5, 7, 12, 15, 17, 18, 22};
		if (!g_zstd_compress)
			return -1;
		if (prm->zstd_level) {
			int st;
			d.zstd_level = prm->zstd_level;
			for (st = 1; st <= 9; st++)
				if (d.zstd_level <= by_level[st]) {
					zstd_strategy = st;
					if (!prm->rzip_level)
						rzip_level = st;
					break;
				}
		} else {
			d.zstd_level = by_level[d.level];
			zstd_strategy = d.level;
		}
	}

	{
		struct plan_out po;
		i64 size_seen = prm->file_size > n ? prm->file_size : n;
		if (prm->stdin_mode) {
			if (make_plan(prm, 0, &po))
				return -1;
			max_chunk = po.max_mmap; /* every pass: chunk_size = mmap_size = max_mmap, src/rzip.c:1046, 1075 */
			size_seen = max_chunk < n ? max_chunk : n;
		}
		if (make_plan(prm, size_seen, &po))
			return -1;
		d.threads = po.threads;
		d.dict_size = po.dict_size;
		d.bufsize = po.bufsize;
		if (!prm->stdin_mode)
			max_chunk = po.max_chunk;
	}
	lfs.stream_bufsize = d.bufsize;
	lfs.threads_used = d.threads;
	lfs.dict_size = d.dict_size;
	if (prm->verbose)
		fprintf(stderr, "oracle: threads %d bufsize %lld dict %u\n", d.threads, (long long)d.bufsize, d.dict_size);

	lrzo_hash_index(hx);
	lrzo_md5_init(&md5);
	pthread_mutex_init(&d.mu, NULL);
	pthread_cond_init(&d.cv, NULL);
	nworkers = prm->workers > 0 ? prm->workers : 1;
	tids = calloc((size_t)nworkers, sizeof(pthread_t));
	for (w = 0; w < nworkers; w++)
		pthread_create(&tids[w], NULL, worker, &d);

	/* chunk loop, src/rzip.c:1041-1186 */
	len = n;
	{
		int pass = 0, stdin_eof = 0;
		/* STDIN: a chunk that fills its buffer does not know it was the last; the next pass reads 0 bytes,
		 * sets eof and writes an empty chunk (mmap_stdin "Empty file" branch, src/rzip.c:821-826) */
		while (!pass || len > 0 || (prm->stdin_mode && !stdin_eof)) {
			i64 offset = n - len, chunk_size = max_chunk < len ? max_chunk : len;
			int bits = 8, cbytes;
			lrzo_sink sink;
			lrzo_rzip_stats rs;

			while (chunk_size >> bits > 0)
				bits++;
			cbytes = bits / 8 + (bits % 8 ? 1 : 0);
			pass++;
			if (prm->stdin_mode && chunk_size < max_chunk)
				stdin_eof = 1; /* short read: control->eof = st->stdin_eof = 1 */

			chunk_sizes = realloc(chunk_sizes, sizeof(i64) * (size_t)(nchunks + 1));
			chunk_cbytes = realloc(chunk_cbytes, sizeof(int) * (size_t)(nchunks + 1));
			chunk_sizes[nchunks] = chunk_size;
			chunk_cbytes[nchunks] = cbytes;

			d.chunk = in + offset;
			d.chunk_no = (int)nchunks;
			d.sbuf[0] = calloc((size_t)d.bufsize, 1);
			d.sbuf[1] = calloc((size_t)d.bufsize, 1);
			d.sblen[0] = d.sblen[1] = 0;
			sink.ctx = &d;
			sink.put0 = sink_put0;
			sink.put1 = sink_put1;
			lrzo_rzip_chunk(d.chunk, chunk_size, rzip_level, cbytes, hx, &victim_round, &sink, &rs, NULL);
			lrzo_md5_update(&md5, d.chunk, (size_t)chunk_size);
			/* close_stream_out: flush stream 0 then stream 1, unconditionally */
			queue_block(&d, 0, 0);
			queue_block(&d, 1, 0);

			lfs.rz.matches += rs.matches;
			lfs.rz.match_bytes += rs.match_bytes;
			lfs.rz.literals += rs.literals;
			lfs.rz.literal_bytes += rs.literal_bytes;
			lfs.rz.tag_hits += rs.tag_hits;
			lfs.rz.tag_misses += rs.tag_misses;
			lfs.rz.inserts += rs.inserts;
			lfs.rz.lookups += rs.lookups;
			nchunks++;
			len -= chunk_size;
		}
	}

	pthread_mutex_lock(&d.mu);
	d.closing = 1;
	pthread_cond_broadcast(&d.cv);
	pthread_mutex_unlock(&d.mu);
	for (w = 0; w < nworkers; w++)
		pthread_join(tids[w], NULL);
	free(tids);

	/* ordered write, src/stream.c:1716-1821 */
	{
		i64 pos = 21, bi = 0, c;
		uchar magic[21];
		ob_reserve(&ob, 1 << 20);
		memset(ob.p, 0, 21);
		ob.len = 21;
		for (c = 0; c < nchunks; c++) {
			int cb = chunk_cbytes[c], j;
			i64 size = chunk_sizes[c] < PAGE ? PAGE : chunk_sizes[c];
			i64 initial_pos, cur_pos = 0, last_head[2];
			uchar hdr[2] = { (uchar)cb, (uchar)(c == nchunks - 1) };
			ob_write_at(&ob, pos, hdr, 2);
			pos += 2;
			ob_val_at(&ob, pos, size, cb);
			pos += cb;
			initial_pos = pos;
			for (j = 0; j < 2; j++) {
				uchar t = CTYPE_NONE;
				last_head[j] = cur_pos + 1 + cb * 2;
				ob_write_at(&ob, initial_pos + cur_pos, &t, 1);
				ob_val_at(&ob, initial_pos + cur_pos + 1, 0, cb);
				ob_val_at(&ob, initial_pos + cur_pos + 1 + cb, 0, cb);
				ob_val_at(&ob, initial_pos + cur_pos + 1 + 2 * cb, 0, cb);
				cur_pos += 1 + cb * 3;
			}
			for (; bi < d.nblocks && d.blocks[bi].chunk_no == c; bi++) {
				struct block *b = &d.blocks[bi];
				uchar t = (uchar)b->c_type;
				ob_val_at(&ob, initial_pos + last_head[b->streamno], cur_pos, cb);
				last_head[b->streamno] = cur_pos + 1 + cb * 2;
				ob_write_at(&ob, initial_pos + cur_pos, &t, 1);
				ob_val_at(&ob, initial_pos + cur_pos + 1, b->c_len, cb);
				ob_val_at(&ob, initial_pos + cur_pos + 1 + cb, b->s_len, cb);
				ob_val_at(&ob, initial_pos + cur_pos + 1 + 2 * cb, 0, cb);
				cur_pos += 1 + cb * 3;
				ob_write_at(&ob, initial_pos + cur_pos, b->buf, b->c_len);
				cur_pos += b->c_len;
				lfs.n_blocks++;
				if (b->c_type == CTYPE_LZMA)
					lfs.blocks_lzma++;
				else
					lfs.blocks_none++;
				free(b->buf);
			}
			pos = initial_pos + cur_pos;
		}
		{
			uchar dig[16];
			lrzo_md5_final(&md5, dig);
			ob_write_at(&ob, pos, dig, 16);
		}
		/* write_magic, src/lrzip.c:131-208 */
		memset(magic, 0, sizeof(magic));
		memcpy(magic, "LRZI", 4);
		magic[4] = 0;
		magic[5] = 14;
		{
			int i;
				/* "else if (control->eof)": to a file the magic is written last (eof set by then); to STDOUT with the
			 * first block of the first chunk, when eof is only set if that chunk is also the last */
			if (!prm->stdout_mode || nchunks == 1)
				for (i = 0; i < 8; i++)
					magic[6 + i] = (uchar)((uint64_t)n >> (8 * i));
		}
		magic[14] = 1; /* MD5 */
		magic[16] = (uchar)prm->filter_flag;
		if (prm->zstd && !prm->no_compress) { /* src/lrzip.c:177-183 */
			magic[17] = (uchar)((zstd_strategy << 4) + 4);
			magic[18] = (uchar)d.zstd_level;
		} else if (lzma_on) {
			magic[17] = 1;
			magic[18] = (uchar)lzma2_prop_from_dic(d.dict_size);
		}
		magic[19] = (uchar)((rzip_level << 4) + d.level);
		ob_write_at(&ob, 0, magic, 21);
	}
	lfs.n_chunks = nchunks;
	free(d.blocks);
	free(chunk_sizes);
	free(chunk_cbytes);
	pthread_mutex_destroy(&d.mu);
	pthread_cond_destroy(&d.cv);
	*out = ob.p;
	*out_len = ob.len;
	if (fs)
		*fs = lfs;
	return 0;
}
