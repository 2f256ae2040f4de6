/* lrzgpu.h -- C ABI of the MI355X-native lrzip-next compression hot path.
 *
 * Drop-in boundary for: rzip long-range preprocessor + lz4 compressibility gate + per-block
 * LZMA backend of lrzip-next 0.14 (reference paths are relative to the lrzip-next tree).
 * Plain pointers and sizes only; every function returns 0 (or a documented value) on success
 * and a negative value on failure -- the library never calls exit().  Host pointers unless the
 * name ends in _dev.  The HIP runtime is required: without a visible gfx950 device the compute
 * entry points fail with LRZGPU_E_NODEVICE (there is no CPU fallback in this library).
 */
#ifndef LRZGPU_H
#define LRZGPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LRZGPU_E_NODEVICE (-100)
#define LRZGPU_E_PARAM (-101)
#define LRZGPU_E_NOMEM (-102)
#define LRZGPU_E_HIP (-103)
#define LRZGPU_E_IO (-104)
#define LRZGPU_E_INTERNAL (-105)
#define LRZGPU_E_FORMAT (-106) /* not a .lrz this library can read, or a failed CRC/MD5/size check */
#define LRZGPU_E_PEER (-107)   /* sharded run: another rank failed (that rank returns its own error) */
#define LRZGPU_E_BLOCK_TOO_LARGE (-108) /* the plan's LZMA block (stream_bufsize) needs a finder workspace beyond this
                                          device: see lrzgpu_max_block_bytes(); returned before any work is done */

/* rzip_control.flags bits this path reads (src/include/lrzip_private.h:257-370) */
#define LRZGPU_FLAG_NO_COMPRESS (1u << 5)  /* FLAG_NO_COMPRESS, -n */
#define LRZGPU_FLAG_THRESHOLD (1u << 20)   /* FLAG_THRESHOLD: lz4 test on (default) */
#define LRZGPU_FLAG_NOBEMT (1u << 27)      /* FLAG_NOBEMT */
#define LRZGPU_FLAG_ZSTD (1u << 26)        /* FLAG_ZSTD_COMPRESS: --zstd back end (host libzstd), GPU rzip + gate */

/* The fields of rzip_control (src/include/lrzip_private.h:472-581) the compress path depends on.
 * Output depends on them exactly as in the reference (block boundaries, dictionary, thread slots:
 * src/stream.c:1169-1331, src/util.c:103-188, src/rzip.c:999-1020). */
typedef struct lrzgpu_control {
	int compression_level;      /* -L, 1..9                                                        */
	int rzip_compression_level; /* -R, 0 = same as compression_level (src/main.c:779-780)          */
	int threads;                /* -p, before prepare_streamout_threads() adds one                 */
	int processors;             /* PROCESSORS (sysconf) as the reference host would report         */
	int64_t ramsize;            /* -m x 100 MiB, or physical RAM (src/lrzip.c:108)                 */
	int64_t window;             /* -w, 0 = unset                                                   */
	uint32_t dictSize;          /* --dictsize, 0 = by level (src/util.c:108-127)                   */
	uint32_t flags;             /* LRZGPU_FLAG_*                                                   */
	int threshold;              /* -T value, default 100                                           */
	/* execution (no influence on the bytes produced) */
	int device;                 /* HIP device ordinal                                              */
	int host_threads;           /* host parser/range-coder threads, 0 = threads                    */
	int gpu_slots;              /* LZMA blocks resident on the GPU at once, 0 = default            */
	int verbose;
	/* results */
	int64_t st_size;            /* control->st_size                                                */
	uint8_t hash_resblock[16];  /* MD5 of the input (control->hash_resblock)                       */
	uint8_t lzma_properties[5]; /* control->lzma_properties                                        */
	uint32_t dictSize_used;     /* dictionary after open_stream_out()'s reduction loop             */
	int64_t stream_bufsize;     /* block size open_stream_out() settled on                         */
	int threads_used;
	/* --zstd only (input, appended here to keep the layout above stable) */
	int zstd_level;             /* --zstd-level 1..22, 0 = from -L (src/main.c:87, 692-711, 822-828) */
	/* execution (no influence on the bytes produced; appended) */
	int scan_slots;             /* rzip chunks scanned concurrently on the GPU, 0 = default (8)      */
	/* stream API only (input, appended) */
	int eof;                    /* control->eof: set before the last chunk's open_stream_out (src/rzip.c:1173-1174) */
	/* per-run selections the reference keeps in rzip_control too (appended; lrzgpu_control_init() sets the defaults) */
	int hash_code;              /* control->hash_code, -H: 0 CRC only, 1 MD5 (default) ... 13, the codes of hashes[]
	                               (src/main.c:64-79); magic[14]; the digest is appended after the last chunk   */
	int filter_flag;            /* control->filter_flag: 0 none, 1 x86, 2 ARM, 3 ARMT, 4 PPC, 5 SPARC, 6 IA64, 7 ARM64,
	                               8 RISC-V, 128 delta; every literal (stream 1) block goes through it before its
	                               back end (src/stream.c:1587-1628), magic[16], and the lz4 test is off
	                               (src/main.c:858-861)                                                           */
	int delta;                  /* control->delta: distance of the delta filter, 1..16, 32, 48 ... 256             */
	int stdin_mode;             /* FLAG_STDIN: fd_in is a stream of unknown length -- chunks are max_mmap bytes each,
	                               an input that ends exactly on a chunk boundary is followed by an empty last chunk,
	                               open_stream_out() sizes the blocks from the first chunk (src/rzip.c:970-973,
	                               1014-1017, 1041-1107, mmap_stdin 800-836)                                     */
	int stdout_mode;            /* FLAG_STDOUT: maxram = ramsize / 6 (src/util.c:179-188) and the magic goes out with the
	                               first chunk, so it carries st_size only when that chunk is also the last one
	                               (src/stream.c:1725-1729, src/lrzip.c:141-144)                                  */
	/* results (appended) */
	uint8_t hash_full[64];      /* the whole digest of hash_code (hash_resblock keeps its first 16 bytes)       */
	int fd_out;                 /* control->fd_out: where lrzgpu_write_1g() / lrzgpu_put_fdout() write (read side)  */
	int backoff_would_apply;    /* out: open_stream_out() probes malloc(limit + overhead x threads) and takes a tenth off
	                               `limit` -- hence the block size -- for as long as the host refuses (retest_malloc,
	                               src/stream.c:1290-1305).  malloc_probe == 0: the blocks are sized as if the first probe
	                               succeeded and this says whether this host would have refused it (1 / 0);
	                               malloc_probe == 1: the number of tenths really taken off                              */
	int malloc_probe;           /* in: 1 = make that probe for real, like the reference (block sizes then depend on what
	                               this process may allocate: address-space rlimit, overcommit policy); 0 (default) =
	                               sizes are a function of the parameters alone                                          */
} lrzgpu_control;

void lrzgpu_control_init(lrzgpu_control *c); /* initialise_control() defaults, src/lrzip.c:1813-1857 */

/* ---- whole-file entry points ----------------------------------------------------------------- */

/* void rzip_fd(rzip_control*, int fd_in, int fd_out)  -- src/include/rzip.h:12, src/rzip.c:922.
 * Compresses fd_in chunk by chunk at the current offset of fd_out and appends the MD5. Like the
 * reference it does not write the 21-byte magic (compress_file does, src/lrzip.c:1464-1560). */
int lrzgpu_rzip_fd(lrzgpu_control *control, int fd_in, int fd_out);

/* compress_file() for plain files: magic placeholder, rzip_fd, write_magic -- src/lrzip.c:1464.
 * Both fd entry points read a regular fd_in chunk by chunk and write every chunk as soon as its blocks are done
 * (host memory holds the blocks in flight, not the file).  control->stdin_mode / stdout_mode reproduce the
 * reference reading STDIN / writing STDOUT byte for byte (chunks of max_mmap bytes cut as mmap_stdin() cuts them,
 * blocks sized from the first chunk, maxram = ramsize / 6 and a size-less magic for STDOUT; src/rzip.c:800-836,
 * 970-973, 1014-1017, 1041-1107, src/util.c:179-188, src/stream.c:1725-1729): the fd is then read as a stream from
 * its current offset.  A pipe as fd_in WITHOUT stdin_mode is spooled and compressed like a regular file of that
 * size.  A non-seekable fd_out gets the image at the end. */
int lrzgpu_compress_file(lrzgpu_control *control, int fd_in, int fd_out);

/* Same container, memory to memory. in: host buffer. *out is malloc'd (caller frees with free()). */
int lrzgpu_compress_buffer(lrzgpu_control *control, const uint8_t *in, int64_t n, uint8_t **out, int64_t *out_len);

/* Same, input already resident in HBM (d_in is a device pointer on control->device). */
int lrzgpu_compress_buffer_dev(lrzgpu_control *control, const void *d_in, int64_t n, uint8_t **out, int64_t *out_len);

/* ---- one file, one process per GPU ---------------------------------------------------------------
 * rzip chunks are independent units of the format (own header, hash table, CRC, block offsets relative to
 * the chunk: src/rzip.c:599-626, src/stream.c:1740-1770), so chunk k of a file goes to GPU k mod G.
 * lrzgpu_compress_chunks* runs the whole path (scan, gate, LZMA blocks) for the chunks k with
 * k % stride == first of the n-byte input and calls on_chunk once per finished chunk, in ascending k,
 * with the chunk's bytes exactly as they stand in the file (chunk header, stream headers, chained
 * blocks).  The one value that crosses a chunk boundary is insert_hash()'s static victim_round
 * (src/rzip.c:308): victim_in[k] >= 0 gives the value chunk k starts from, victim_in == NULL or a
 * negative entry lets the library predict it (what chunk k-1 left when this call scanned it, else 0);
 * on_chunk reports the value used and the value left, so the caller that owns the file can check the
 * chain victim_out[k-1] == victim_in[k] and ask again for a chunk whose guess was wrong: what on_chunk reports
 * is authoritative -- with stride > 1 the library cannot check the chain itself (it does with stride == 1).
 * with_md5: also compute the MD5 of the whole input into control->hash_resblock (one caller does).
 * lrzgpu_assemble_chunks (host only): magic + the chunk images in order + the whole-input hash = the .lrz file;
 * digest = lrzgpu_hash_length(control->hash_code) bytes (16 for MD5, up to 64; none read for code 0). */
typedef int (*lrzgpu_chunk_fn)(void *ctx, int chunk_index, int64_t victim_in, int64_t victim_out,
			       const uint8_t *chunk_image, int64_t len);
int lrzgpu_compress_chunks(lrzgpu_control *control, const uint8_t *in, int64_t n, int first, int stride,
			   const int64_t *victim_in, int with_md5, lrzgpu_chunk_fn on_chunk, void *ctx);
int lrzgpu_compress_chunks_dev(lrzgpu_control *control, const void *d_in, int64_t n, int first, int stride,
			       const int64_t *victim_in, int with_md5, lrzgpu_chunk_fn on_chunk, void *ctx);
int lrzgpu_assemble_chunks(lrzgpu_control *control, int64_t st_size, int n_chunks, const uint8_t *const *chunk_img,
			   const int64_t *chunk_len, const uint8_t *digest, uint8_t **out, int64_t *out_len);

/* The whole one-file-over-N-ranks protocol behind the C ABI (csrc/shard.cpp): every rank calls the same function with
 * its rank / world and three transport callbacks; rank r compresses chunks r, r + world, ... (the whole GPU path),
 * the ranks agree on the victim_round chain (one all-reduce of three integers per chunk per round; the owner of the
 * first chunk whose guess was wrong redoes it), the chunk images go to rank 0 in file order (send / recv straight out
 * of pinned host memory: the chunk hand-off), rank 0 returns the .lrz (*out malloc()ed; NULL on the other ranks).
 * No collective touches the data path.  The transport is the caller's: RCCL over xGMI (bench.py wraps
 * torch.distributed), gloo (tests/test_sharded_cpu.py), MPI ...; callbacks return 0 on success.
 *   lrzgpu_compress_sharded_dev         the whole input resident on every rank's device (rank 0 hashes it on its own
 *                                        thread from the start, beside the chunks)
 *   lrzgpu_compress_sharded_chunks_dev  a rank holds only ITS chunks: d_chunks[k] = device pointer of chunk k for
 *                                        k % world == rank (chunk sizes: lrzgpu_plan), NULL elsewhere; rank 0 cannot
 *                                        use this form unless control->hash_code == 0 (the hash needs every byte)
 *   lrzgpu_shard_protocol               the protocol alone over any per-rank chunk compressor `fn` (same contract as
 *                                        lrzgpu_compress_chunks: chunks k % stride == first, victim_in[k] >= 0 = start
 *                                        value, on_chunk per finished chunk); digest = the whole-input hash (rank 0)
 * *redone (may be NULL): chunks compressed again because of the chain, over all ranks.
 * Failure: a rank whose compressor fails (out of memory, a HIP error) still enters the next all-reduce with an error
 * flag set, so every rank leaves with an error together -- the failing rank with its own code, the others with
 * LRZGPU_E_PEER -- before any send / recv.  A failing TRANSPORT cannot be reported through itself: the callbacks must be
 * abortable (a failed or timed-out collective / send / recv on one rank has to fail the peers' pending calls:
 * ncclCommAbort, a gloo timeout), and any callback error ends the protocol with LRZGPU_E_IO on that rank. */
typedef struct lrzgpu_shard_comm {
	void *ctx;
	int rank, world;
	int (*allreduce_sum_i64)(void *ctx, int64_t *vals, int count);   /* in place, over all ranks            */
	int (*send)(void *ctx, int dst, const void *buf, int64_t n);     /* blocking, host memory (pinned)      */
	int (*recv)(void *ctx, int src, void *buf, int64_t n);
} lrzgpu_shard_comm;
typedef int (*lrzgpu_shard_compress_fn)(void *ctx, int first, int stride, const int64_t *victim_in, lrzgpu_chunk_fn on_chunk,
					void *on_chunk_ctx);
int lrzgpu_compress_sharded_dev(lrzgpu_control *control, const void *d_in, int64_t n, const lrzgpu_shard_comm *comm,
				uint8_t **out, int64_t *out_len, int64_t *redone);
int lrzgpu_compress_sharded_chunks_dev(lrzgpu_control *control, const void *const *d_chunks, int64_t n,
				       const lrzgpu_shard_comm *comm, uint8_t **out, int64_t *out_len, int64_t *redone);
int lrzgpu_compress_sharded(lrzgpu_control *control, const uint8_t *in, int64_t n, const lrzgpu_shard_comm *comm,
			    uint8_t **out, int64_t *out_len, int64_t *redone);
int lrzgpu_shard_protocol(lrzgpu_control *control, int64_t n, const lrzgpu_shard_comm *comm, lrzgpu_shard_compress_fn fn,
			  void *fn_ctx, const uint8_t *digest, uint8_t **out, int64_t *out_len, int64_t *redone);

/* The transport in C over RCCL (csrc/shard_rccl.cpp) -- xGMI between the GPUs of a node; "RCCL over xGMI only for chunk
 * hand-off" as the path's definition says.  lrzgpu_rccl_comm_create fills a lrzgpu_shard_comm whose three callbacks
 * are ncclAllReduce on a device copy of the words and ncclSend / ncclRecv through a pair of 32 MiB staging buffers
 * (pinned host -> staging on a copy stream beside the send of the piece before; the receive of the next piece queued
 * before a piece is copied out).  Bootstrap as with RCCL itself: rank 0 makes the 128-byte id, the caller carries it to
 * the other ranks (MPI_Bcast, a file, a TCP store), every rank creates its communicator on its own device.  RCCL is
 * taken from the process at run time (dlopen librccl.so.1); without it these return LRZGPU_E_NODEVICE.  A failed call
 * aborts the communicator (ncclCommAbort), which fails the peers' pending calls: the abortable transport the protocol
 * asks for; a wait for a peer that never shows up gives up the same way: after ten minutes in a send or a receive,
 * after an hour in the all-reduce (where a rank that is through waits for the slowest one to compress, or redo, a chunk).  lrzgpu_rccl_loopback: self test of the staging / send / receive path on one rank (grouped send + receive to
 * itself). */
#define LRZGPU_RCCL_ID_BYTES 128
int lrzgpu_rccl_available(void);
/* Take ncclGetUniqueId / CommInitRank / AllReduce / Send / Recv / Group* / CommAbort / CommDestroy from this shared object
 * instead of librccl.so.1 (an RCCL build under another name; tests/stubs/nccl_stub.cpp: an in-process stand-in with which
 * two ranks are two threads on one GPU).  Only before the first other lrzgpu_rccl_* call of the process. */
int lrzgpu_rccl_use_library(const char *path);
int lrzgpu_rccl_unique_id(uint8_t id[LRZGPU_RCCL_ID_BYTES]);
int lrzgpu_rccl_comm_create(const uint8_t id[LRZGPU_RCCL_ID_BYTES], int rank, int world, int device, lrzgpu_shard_comm *comm);
int lrzgpu_rccl_comm_destroy(lrzgpu_shard_comm *comm);
int lrzgpu_rccl_loopback(lrzgpu_shard_comm *comm, const void *src, void *dst, int64_t n);

/* ---- stream layer, compress side: src/include/stream.h:14-32 kept call for call ---------------------
 * For a caller that produces the two rzip streams itself (the reference's hash_search() through
 * put_header/put_literal/put_match, src/rzip.c:184-265): the same functions with the same argument lists,
 * `rzip_control` -> `lrzgpu_control`, bool/void -> int (0 = ok, negative = error instead of fatal()).
 *   prepare_streamout_threads  src/stream.c:1090-1118   ring of threads + 1 block slots (1 under -n)
 *   open_stream_out            src/stream.c:1140-1348   one per chunk; the first call fixes threads /
 *                              dictionary / stream_bufsize from control->st_size, -p, -m (set st_size first);
 *                              control->eof is latched for the chunk header; returns the sinfo handle
 *   write_stream               src/stream.c:2198-2216   append to stream 0 (tokens) or 1 (literals); a full
 *                              buffer is handed to the back end (lz4 gate, GPU finder, host parser)
 *   flush_buffer               src/stream.c:1878-1881   hand the current (partial) buffer off now
 *   close_stream_out           src/stream.c:2253-2282   flush stream 0 then 1, empty blocks included
 *   close_streamout_threads    src/stream.c:1121-1136   wait for every block; leaves fd at the end of the
 *                              last chunk (the caller appends the hash and rewrites the magic)
 * Blocks reach the file strictly in hand-off order with the chunk header in front of a chunk's first
 * block (compthread, src/stream.c:1716-1821).  The output fd must be seekable. */
int lrzgpu_prepare_streamout_threads(lrzgpu_control *control);
int lrzgpu_close_streamout_threads(lrzgpu_control *control);
void *lrzgpu_open_stream_out(lrzgpu_control *control, int f, unsigned int n, int64_t chunk_limit, char cbytes);
int lrzgpu_flush_buffer(lrzgpu_control *control, void *sinfo, int stream);
int lrzgpu_write_stream(lrzgpu_control *control, void *ss, int streamno, const uint8_t *p, int64_t len);
int lrzgpu_close_stream_out(lrzgpu_control *control, void *ss);

/* ---- stream layer, read side: src/include/stream.h:20-21, 26, 29, 31, 32 -----------------------------------
 * For a caller that replays the rzip tokens itself, the way runzip_chunk() does (src/runzip.c:139-370): it reads the
 * chunk_bytes byte of a chunk with read_1g, opens the chunk's two streams, pulls token headers / match offsets from
 * stream 0 and literal bytes from stream 1, writes what it reconstructs with write_1g, and closes the chunk.
 *   open_stream_in   src/stream.c:1352-1506  f stands right after the chunk_bytes byte; reads the eof flag into
 *                    control->eof, adds the chunk size to control->st_size, checks the two initial stream headers;
 *                    blocks are then fetched and decoded ahead of the reader by worker threads (fill_buffer
 *                    2023-2195, ucompthread 1883-2021: stored / LZMA / zstd blocks; the filter named by
 *                    control->filter_flag / delta -- the caller takes them from the magic, lrzgpu_read_magic -- is
 *                    undone on every literal block)
 *   read_stream      src/stream.c:2220-2250  bytes read (fewer than asked for at the end of the stream), -1 on failure
 *   close_stream_in  src/stream.c:2299-2319  leaves f where the next chunk, or the hash, starts; frees the handle
 *   write_1g / put_fdout   src/stream.c:802-850   to control->fd_out;  read_1g  src/stream.c:897-945
 *   get_readseek     src/stream.c:1078-1088  (src/include/stream.h:22; runzip_chunk calls it, src/runzip.c:293): the
 *                    offset fd stands at, -1 if it cannot be told (the reference calls fatal() there) */
void *lrzgpu_open_stream_in(lrzgpu_control *control, int f, int n, char cbytes);
int64_t lrzgpu_read_stream(lrzgpu_control *control, void *ss, int streamno, uint8_t *p, int64_t len);
int lrzgpu_close_stream_in(lrzgpu_control *control, void *ss);
int64_t lrzgpu_write_1g(lrzgpu_control *control, const void *buf, int64_t len);
int64_t lrzgpu_read_1g(lrzgpu_control *control, int fd, void *buf, int64_t len);
int64_t lrzgpu_put_fdout(lrzgpu_control *control, const void *offset_buf, int64_t ret);
int64_t lrzgpu_get_readseek(lrzgpu_control *control, int fd);

/* The per-block back-end dispatch seam: static int lzma_compress_buf(rzip_control*, struct compress_thread*,
 * int current_thread), src/stream.c:429-494 (and zstd_compress_buf 167-230 under LRZGPU_FLAG_ZSTD), called
 * by compthread (src/stream.c:1633-1648).  Same contract: s_buf is malloc()ed and owned by *cthread; on
 * success with a smaller result the old s_buf is freed, the new one installed, c_len / c_type (6 LZMA,
 * 10 zstd) set; "left uncompressed" (lz4 gate said no, incompressible, does not fit) also returns 0 with
 * *cthread untouched; -1 = a resource was missing -- the reference's caller then waits for its predecessor
 * and tries once more (src/stream.c:1667-1714).  SZ_ERROR_MEM lowers control->compression_level and retries
 * like the reference (src/stream.c:462-467).  The calling thread keeps its device buffers between calls. */
typedef struct lrzgpu_compress_thread { /* struct compress_thread, src/stream.c:67-76 (no semaphore/salt) */
	uint8_t *s_buf;   /* uncompressed buffer -> compressed buffer */
	uint8_t c_type;   /* 3 = CTYPE_NONE on entry */
	int64_t s_len;    /* data length uncompressed */
	int64_t c_len;    /* data length compressed (= s_len on entry) */
	void *sinfo;
	int streamno;
} lrzgpu_compress_thread;
int lrzgpu_lzma_compress_buf(lrzgpu_control *control, lrzgpu_compress_thread *cthread, int current_thread);

/* ---- rzip stage -------------------------------------------------------------------------------
 * hash_search(), src/rzip.c:586-762: scan one chunk resident in HBM.  Emits the two rzip streams
 * exactly as put_match/put_literal/write_sbstream would (src/rzip.c:208-265): stream 0 (tokens,
 * terminator, CRC) into a malloc'd host buffer, stream 1 (literal bytes) into a device buffer of
 * the caller (capacity n). victim_round carries insert_hash()'s static (src/rzip.c:308). */
typedef struct lrzgpu_scan_stats {
	int64_t matches, match_bytes, literals, literal_bytes, inserts, lookups, tag_hits, tag_misses;
	int64_t hash_count, tag_clean_ptr;
	uint64_t minimum_tag_mask, tag_mask;
} lrzgpu_scan_stats;

int lrzgpu_hash_search_dev(const void *d_chunk, int64_t chunk_size, int rzip_level, int chunk_bytes,
			   int64_t *victim_round, uint8_t **stream0, int64_t *stream0_len,
			   void *d_stream1, int64_t *stream1_len, uint32_t *crc32, lrzgpu_scan_stats *stats,
			   int device);
/* host-pointer convenience wrapper (uploads the chunk, downloads stream 1 into stream1[n]) */
int lrzgpu_hash_search(const uint8_t *chunk, int64_t chunk_size, int rzip_level, int chunk_bytes,
		       int64_t *victim_round, uint8_t **stream0, int64_t *stream0_len,
		       uint8_t *stream1, int64_t *stream1_len, uint32_t *crc32, lrzgpu_scan_stats *stats,
		       int device);
/* init_hash_indexes(), src/rzip.c:765-771 (glibc random() seed-1 sequence, frozen) */
void lrzgpu_hash_index(uint64_t out[256]);

/* The scan's duplicate census on its own (csrc/rzip_census.hip): 1 = no 31-byte window of s_buf[0..s_len) occurs twice --
 * exactly: no rzip match (MINIMUM_MATCH 31, src/rzip.c:431-461) can exist in such a chunk, and the whole-file entry
 * points do not run the table automaton on the file's last chunk then --, 0 = some may.  stats (may be NULL): anchors
 * and equal values of the 1/64 sample, anchors and equal values of the full pass, and how many of the latter were chance
 * (equal 8-byte values whose surroundings differ: looked at, cleared). */
int lrzgpu_census(const uint8_t *s_buf, int64_t s_len, int device, int64_t stats[5]);

/* ---- lz4 gate ---------------------------------------------------------------------------------
 * static int lz4_compresses(rzip_control*, uchar *s_buf, i64 s_len) -- src/stream.c:2325-2380.
 * Returns 0 = leave the block uncompressed, 1..100 = percentage; <0 on error. */
int lrzgpu_lz4_compresses(const uint8_t *s_buf, int64_t s_len, int threshold, int device);
int lrzgpu_lz4_compresses_dev(const void *d_buf, int64_t s_len, int threshold, int device);
/* LZ4_compress_default(src, dst, srcSize, dstCapacity) return value (liblz4 1.9.3), size only. */
int lrzgpu_lz4_compress_default_size(const uint8_t *src, int src_size, int dst_capacity, int device);
/* The same with the early verdict the pipeline uses: the exact size, or -- as soon as "size <
 * stop_below" is certain (what is left costs at most rest + rest/255 + 16 bytes) -- an upper bound of
 * the size that is itself below stop_below. */
int lrzgpu_lz4_size_stop_below(const uint8_t *src, int src_size, int dst_capacity, int stop_below, int device);

/* ---- LZMA backend -----------------------------------------------------------------------------
 * LzmaCompress() -- src/lzma/include/LzmaLib.h:95-112, same arguments and SRes codes
 * (0 OK, 2 MEM, 5 PARAM, 7 OUTPUT_EOF).  GPU match finder + host parser/range coder.
 * Levels 5..9: BT4 finder (btMode=1, numHashBytes=4, as the reference's MT finder returns it) + optimal
 * parser; levels 1..4: HC5 finder (btMode=0, numHashBytes=5, LzFind.c:1431-1502) + GetOptimumFast
 * (LzmaEnc.c:1970-2098).  numThreads is accepted and ignored (the output of the reference does not
 * depend on it). */
int lrzgpu_LzmaCompress(unsigned char *dest, size_t *destLen, const unsigned char *src, size_t srcLen,
			unsigned char *outProps, size_t *outPropsSize, int level, unsigned dictSize,
			int lc, int lp, int pb, int fb, int numThreads);

/* The GPU half alone: per-position match lists in the order MatchFinderMt_GetMatches
 * (src/lzma/C/LzFindMt.c:1274-1317) would return them.  counts[i] = number of u32 entries of
 * position i; pairs = (len, dist-1) couples of position 0, 1, ... ; returns total entries or <0. */
int64_t lrzgpu_lzma_match_lists(const uint8_t *src, size_t n, uint32_t dictSize, unsigned fb, unsigned cutValue,
				uint8_t *counts, uint32_t *pairs, size_t pairs_cap, int device);
/* The BT4 finder on a PREFIX src[0..n) of a block of block_n bytes (the early start of a block): the
 * hash mask is the block's, so the lists of the positions below n - fb - 4 are the whole block's; the last fb + 4
 * positions' lists are clipped by the end of the prefix and must not be used. */
int64_t lrzgpu_lzma_match_lists_prefix(const uint8_t *src, size_t n, size_t block_n, uint32_t dictSize, unsigned fb,
				       unsigned cutValue, uint8_t *counts, uint32_t *pairs, size_t pairs_cap, int device);
/* Same for the hash-chain finder of levels 1..4: the lists Hc5_MatchFinder_GetMatches
 * (src/lzma/C/LzFind.c:1431-1502) returns position by position. */
int64_t lrzgpu_lzma_match_lists_hc5(const uint8_t *src, size_t n, uint32_t dictSize, unsigned fb, unsigned cutValue,
				    uint8_t *counts, uint32_t *pairs, size_t pairs_cap, int device);

/* The same finder as a stream of blocks in the BT thread's format -- what BtGetMatches() hands to the LZ thread
 * (src/lzma/C/LzFindMt.c:39-42, 571-729; consumer MatchFinderMt_GetNextBlock_Bt 946-981): d[0] = u32 words used
 * in the block including these two, d[1] = bytes available at the block's first position, then per position a
 * record [num, (len, dist-1) x num/2] of the binary-tree matches (length >= 4; the h2/h3 candidates are the LZ
 * thread's own business, MixMatches3), num = 0 for none, [2, len, dist-1] inside long runs.  A block is filled
 * while fewer than 65536 - 2 * fb words are used, like the reference's; where exactly the reference cuts its
 * blocks also depends on its hash-thread phases, which the consumer does not see.  open() runs the finder for the
 * whole block of data at once (lists are independent of the parser's decisions); next_block() returns the number
 * of positions it put into btBuf (capacity >= 65536 words), 0 after the last one, < 0 on error. */
typedef struct lrzgpu_mf lrzgpu_mf;
int lrzgpu_lzma_mf_open(lrzgpu_mf **mf, const uint8_t *src, size_t n, uint32_t dictSize, unsigned fb, unsigned cutValue, int device);
int lrzgpu_lzma_mf_next_block(lrzgpu_mf *mf, uint32_t *btBuf, size_t cap_u32);
void lrzgpu_lzma_mf_close(lrzgpu_mf *mf);

/* The host half alone: parser + range coder fed with match lists (any producer). */
int lrzgpu_lzma_encode_with_lists(unsigned char *dest, size_t *destLen, const unsigned char *src, size_t srcLen,
				  const uint8_t *counts, const uint32_t *pairs, int level, unsigned dictSize,
				  int lc, int lp, int pb, int fb);

/* The same with the list formats the pipeline moves over PCIe (lzma_mf.hip k_gather): 0 = the couples above;
 * 1 = bit 31 of every len word carries the finder's tail flag ("after this match and one literal byte the next
 * two bytes continue at the same distance"); 2 = one u32 per pair, flag << 31 | (len - 2) << 25 | dist-1
 * (dictSize <= 32 MiB, fb <= 65).  counts[] counts two entries per pair in every format. */
int lrzgpu_lzma_encode_with_lists_fmt(unsigned char *dest, size_t *destLen, const unsigned char *src, size_t srcLen,
				      const uint8_t *counts, const uint32_t *pairs, int list_format, int level,
				      unsigned dictSize, int lc, int lp, int pb, int fb);

/* The same on lists that arrive in stages -- the early start of a block whose tail is still being scanned
 * (csrc/lzma_enc.h StagedLists is the in-library interface, the whole-file driver its user).
 * This entry is its host-only harness: the encoder runs on a private copy of src[] whose bytes from the current limit
 * on are overwritten and on lists cut off there; each time it asks for more, stage_step further positions are
 * revealed (0 = everything that is left, in one piece); the stream must equal lrzgpu_lzma_encode_with_lists_fmt()'s.
 * early_counts / early_pairs (may be NULL: the whole block's lists are used) = the lists of the first
 * early_positions positions as a finder run on a prefix of the block produced them. */
int lrzgpu_lzma_encode_with_lists_staged(unsigned char *dest, size_t *destLen, const unsigned char *src, size_t srcLen,
					 const uint8_t *counts, const uint32_t *pairs, size_t early_positions, int list_format,
					 int level, unsigned dictSize, int lc, int lp, int pb, int fb,
					 const uint8_t *early_counts, const uint32_t *early_pairs, size_t stage_step);

/* ---- host-only pieces of the stream layer (usable without a device) ----------------------------
 * lrzgpu_plan: the sizing open_stream_out()/rzip_fd() derive before the first chunk
 * (src/stream.c:1169-1331, src/rzip.c:999-1020); fills stream_bufsize, dictSize_used, threads_used.
 * lrzgpu_container_store: lays out a .lrz from ready-made rzip streams with every block stored
 * (CTYPE_NONE), i.e. what compthread writes under -n (src/stream.c:1716-1821) plus write_magic. */
int lrzgpu_plan(lrzgpu_control *control, int64_t st_size, int64_t *chunk_size);
int lrzgpu_container_store(lrzgpu_control *control, int64_t st_size, int n_chunks, const int64_t *chunk_sizes,
			   const uint8_t *const *stream0, const int64_t *stream0_len,
			   const uint8_t *const *stream1, const int64_t *stream1_len, const uint8_t md5[16],
			   uint8_t **out, int64_t *out_len);

/* ---- measurement --------------------------------------------------------------------------------
 * Per-kernel durations measured with HIP events on the stream each kernel is launched on, summed
 * over launches since the last reset, plus the algorithmic work counters they processed. */
typedef struct lrzgpu_profile {
	double tag_scan_ms, resolve_ms, crc_ms, gather_ms, lz4_ms, mf_bt_ms, mf_total_ms;
	int64_t tag_scan_launches, resolve_launches, crc_launches, gather_launches, lz4_launches, mf_launches;
	int64_t tag_scan_positions;   /* positions tagged by k_tag_scan                       */
	int64_t resolve_lookups;      /* candidates probed by k_resolve                       */
	int64_t resolve_inserts;
	int64_t resolve_match_bytes;  /* bytes covered by emitted matches (verified both sides) */
	int64_t crc_bytes, gather_bytes, lz4_bytes;
	int64_t mf_positions;         /* block bytes through the match finder                 */
	int64_t mf_entries;           /* u32 match-list entries produced                      */
	double scan_wall_ms;          /* host wall time inside scan_chunk_device              */
	int64_t resolve_dbg[16];      /* batches, committed lanes, serial steps, stops: complex, match,
	                                 conflict, no-victim, swept-range; [8..13] shader cycles in
	                                 refill, simulate, victim scan, conflict test, apply, tail
	                                 (cycle laps only with LRZGPU_RESOLVE_PROF=1)              */
	double long_compare_ms;       /* k_long_compare: forward extents of matches > 4 MiB    */
	int64_t long_compare_launches;
	int64_t long_compare_bytes;   /* bytes of both operands it compared                    */
	int64_t spec_rollbacks;       /* chunks whose early-released literal frontier a later match crossed
	                                 (stream 1 rebuilt, early blocks discarded)              */
	int64_t spec_cancelled_blocks; /* early-released blocks thrown away by those roll-backs   */
	int64_t victim_rescans;       /* chunks scanned again because the victim_round they started from
	                                 was not what their predecessor left (src/rzip.c:308)        */
	/* Launches of one kind run side by side on different streams (a resolver per chunk, a finder per GPU slot), so
	 * the summed durations above exceed the wall time.  Per kind -- 0 k_tag_scan, 1 k_resolve, 2 k_crc32_tiles,
	 * 3 k_gather_runs, 4 k_lz4_size, 5 k_bt (+ k_bt_wave), 6 the whole finder, 7 k_long_compare -- the wall time the
	 * launches cover (union of their [start, end) intervals since lrzgpu_profile_reset) and the most that overlapped. */
	double union_ms[8];
	double peak_concurrency[8];
	/* the whole-file pipeline, summed over the runs since the reset (seconds): 0 host encoders busy (parser + range
	 * coder, all threads), 1 host encoders waiting for a block, 2 GPU workers in the finder, 3 GPU workers copying
	 * lists to the host, 4 when the last chunk's scan ended, 5 when the last finder ended, 6 when the last encoder
	 * ended (4-6: since the start of their run), 7 wall time of the runs */
	double pipeline_s[8];
	int64_t mf_wave_dbg[4];       /* k_bt_wave: rounds of all waves, node visits, rounds walks waited for a son, positions */
	/* early start of blocks (the encoder follows the finder through a block that is still being scanned), summed over
	 * the runs since the reset: 0 when the first encoder started (seconds since the start of its run), 1 encoder seconds
	 * spent waiting for the next part of a started block (counted in pipeline_s[1] too), 2 blocks started early,
	 * 3 finder runs on their prefixes */
	double early_s[4];
	/* one file over N ranks (lrzgpu_compress_sharded*, lrzgpu_shard_protocol), this rank, summed over the runs since the
	 * reset (seconds): 0 this rank's own chunks through the whole path (on rank 0 the whole-input hash runs beside them
	 * and is waited for), 1 the chain check (all-reduce: includes waiting for the slowest rank), 2 chunks redone + their
	 * checks, 3 the hand-off (sending this rank's images / receiving the others' and laying the file out), 4 when the
	 * whole-input hash was done (since the start of its run; rank 0), 5 wall time of the protocol */
	double shard_s[6];
} lrzgpu_profile;
void lrzgpu_profile_reset(void);
void lrzgpu_profile_get(lrzgpu_profile *out);
/* the launch intervals of one kind (the order of union_ms[]): [start, end) pairs in ms since the reset; returns how
 * many exist and writes at most `cap` pairs (tools/timeline.py draws them) */
int lrzgpu_profile_intervals(int kind, double *out, int cap);

/* ---- round-trip verifier ----------------------------------------------------------------------
 * The read side of the same subset of the format (lrzip-next 0.14 magic, stored + LZMA blocks, MD5 or
 * no hash, no encryption/filters): container walk (src/stream.c:1352-1506, 2023-2195), LzmaDec,
 * runzip token replay (src/runzip.c:146-260), chunk CRC and MD5 checks (src/runzip.c:352-440).
 * Host code -- decompression is outside the accelerated path; it lets a GPU box check
 * decode(compress(x)) == x through this ABI.  *out is malloc()ed. */
int lrzgpu_decompress_buffer(const uint8_t *lrz, int64_t n, uint8_t **out, int64_t *out_len, int host_threads);
/* decompress_file() for that subset (src/lrzip.c decompress_file -> runzip_fd): whole fd_in -> fd_out */
int lrzgpu_decompress_file(int fd_in, int fd_out, int host_threads);

/* The figures `lrzip-next -i` prints (get_fileinfo, src/lrzip.c:1069-1460) for an image in memory. */
typedef struct lrzgpu_info {
	int major, minor;          /* 0.14 */
	int64_t st_size;           /* uncompressed size from the magic */
	int64_t compressed_size;
	int hash_code;             /* 0 CRC only, 1 MD5 */
	int lzma, dict_prop;       /* LZMA back end flag and its lzma2-style dictionary byte */
	int level, rzip_level;
	int64_t chunks, blocks, blocks_lzma;
	int64_t stream_c_len[2], stream_u_len[2]; /* stream 0 (tokens) / stream 1 (literals): stored and original bytes */
} lrzgpu_info;
int lrzgpu_file_info(const uint8_t *lrz, int64_t n, lrzgpu_info *info);

/* ---- misc ------------------------------------------------------------------------------------- */
/* Device workspaces, pinned and pageable host buffers are cached between calls (a compressor handles
 * block after block, file after file); this returns all of it to the system. */
void lrzgpu_trim(void); /* parked buffers, workspaces and streams back to the system */
int lrzgpu_device_count(void);
/* The largest LZMA block (stream_bufsize of lrzgpu_plan()) a run on `device` can take: the match finder keeps ~142 B per
 * block byte resident (the reference's: 11.5 B per dictionary byte on the host, src/util.c:108-131).  ~1.9 GB on a 288 GB
 * part; the reference sizes a block as max(limit, overhead - dict) / threads (src/stream.c:1316-1323): -p1 / -p2 on a
 * multi-GB file can exceed that, and such a run returns LRZGPU_E_BLOCK_TOO_LARGE at once. */
int64_t lrzgpu_max_block_bytes(int device);
const char *lrzgpu_version(void);

#ifdef __cplusplus
}
#endif
#endif
