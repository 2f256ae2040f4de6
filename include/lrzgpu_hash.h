/* lrzgpu_hash.h -- container / front-end completeness: the whole-file hashes a .lrz may carry and
 * the magic headers of every archive version the reference still reads.  Host-only entry points of liblrzgpu.so;
 * kept apart from lrzgpu.h, which the device code includes. */
#ifndef LRZGPU_HASH_H
#define LRZGPU_HASH_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* ---- hashes: `hashes[]` of src/main.c:64-79 (code = magic[14]; libgcrypt in the reference, src/rzip.c:943-950,
 * 1195-1219; checked on the read side in src/runzip.c:352-440) -------------------------------------------------
 * 0 CRC (4 bytes, nothing appended to the file), 1 MD5 (16), 2 RIPEMD-160 (20), 3 SHA-256 (32), 4 SHA-384 (48),
 * 5 SHA-512 (64), 6 SHA3-256 (32), 7 SHA3-512 (64), 8-10 SHAKE128 with 16/32/64 bytes, 11-13 SHAKE256 with 16/32/64. */
#define LRZGPU_HASH_MAX 13
int lrzgpu_hash_length(int hash_code);        /* bytes, -1 for an unknown code */
const char *lrzgpu_hash_label(int hash_code); /* "MD5", "SHA3_256", ... as the reference prints them */
/* one shot; out must hold lrzgpu_hash_length(hash_code) (<= 64) bytes.  0 or LRZGPU_E_PARAM */
int lrzgpu_hash_buffer(int hash_code, const uint8_t *data, int64_t n, uint8_t *out);
/* streaming (the reference feeds its hash chunk by chunk from cksumthread, src/rzip.c:564-584) */
void *lrzgpu_hash_open(int hash_code);
int lrzgpu_hash_update(void *h, const uint8_t *data, int64_t n);
int lrzgpu_hash_final(void *h, uint8_t *out); /* also closes h */

/* The hash of a run is control->hash_code (lrzgpu.h), 1 = MD5 unless changed: the whole-file entry points compute
 * and append it; the chunk-sharded ones compute it when asked to (with_md5) and lrzgpu_assemble_chunks appends
 * lrzgpu_hash_length(control->hash_code) bytes.  control->hash_full receives the digest, hash_resblock its first 16
 * bytes.  (-H <n> of the reference's command line = that field; lrzgpu_control_init() sets 1.) */

/* This rewrites the trailer of an image (or
 * of any 0.14 image without encryption) for another hash code: magic[14] = hash_code, the old digest dropped,
 * `digest` (lrzgpu_hash_length(hash_code) bytes of the UNCOMPRESSED data; ignored for code 0) appended.
 * *out is malloc()ed. */
int lrzgpu_set_file_hash(const uint8_t *lrz, int64_t n, int hash_code, const uint8_t *digest, uint8_t **out, int64_t *out_len);

/* ---- read_magic()/get_magic() of src/lrzip.c:262-585 for archive versions 0.6 - 0.14 ------------------------------ */
typedef struct lrzgpu_magic {
	int major, minor;
	int magic_len;          /* bytes of the header proper: 24 (0.6, 0.7), 18 (0.8), 20 (0.9, 0.10), 21 (0.11+) */
	int64_t st_size;        /* expected uncompressed size; 0 when encrypted (the field holds the salt then) */
	int enc_code;           /* 0 none, 1 AES128, 2 AES256 */
	uint8_t salt[8];
	int costfactor;         /* salt[0] */
	int hash_code, hash_len; /* 0 / 0: chunk CRCs only */
	int filter_flag;        /* 0 none, 1 x86, 2 ARM, 3 ARMT, 4 PPC, 5 SPARC, 6 IA64, 7 ARM64, 8 RISC-V, 128 delta */
	int delta;              /* delta distance when filter_flag == 128 */
	int ctype;              /* 0.11+: 0 none, 1 lzma, 2 zpaq, 3 bzip3, 4 zstd; older: 1 when lzma properties are stored, else 0 */
	uint32_t dict_size;     /* lzma */
	uint8_t lzma_properties[5];
	int zpaq_bs, zpaq_level, bzip3_bs, zstd_strategy, zstd_level;
	int level, rzip_level;  /* 0.9+ */
	int comment_length;     /* 0.9+; the comment itself follows the header */
	char comment[256];
} lrzgpu_magic;
/* 0, LRZGPU_E_FORMAT (not an lrzip file / unsupported version / invalid compression type) or LRZGPU_E_PARAM */
int lrzgpu_read_magic(const uint8_t *lrz, int64_t n, lrzgpu_magic *m);

/* ---- filters on the literal stream (src/stream.c:1587-1628 before the back end, 1926-1990 after it;
 * converters: src/lzma/C/Bra.c, Bra86.c, Delta.c) -- host implementations, checked against the reference's own
 * converters; the read side (lrzgpu_decompress_*) undoes them on every stream-1 block.  filter_flag as in magic[16]:
 * 1 x86, 2 ARM, 3 ARMT, 4 PPC, 5 SPARC, 6 IA64, 7 ARM64, 8 RISC-V, 128 delta with distance `delta`
 * (1..16, 32, 48 ... 256). */
/* The filter of a run is control->filter_flag / control->delta (lrzgpu.h), as in the reference's rzip_control: every
 * literal block goes through it before its back end, magic[16] says so, the lz4 test is off (src/main.c:858-861).
 * In the whole-file and chunk-sharded entry points the block is filtered in HBM where the scan left it
 * (lrzgpu_filter_block_dev below); the stream API filters its host buffers with the host converters.
 * (--x86 ... --delta=N of the reference's command line, src/main.c:612-660, = those fields; lrzgpu_control_init()
 * sets none.)  A run reads nothing but its control. */
/* the compress direction over a block resident on `device` (d_data 4-byte aligned), in place: one thread per
 * word / Thumb pair / IA-64 bundle; x86 and RISC-V as candidate compaction + per-run resolution + parallel conversion
 * (csrc/filters_gpu.hip); bit-identical to lrzgpu_filter_block(..., encode = 1) */
int lrzgpu_filter_block_dev(int filter_flag, int delta, void *d_data, int64_t n, int device);
int lrzgpu_filter_supported(int filter_flag, int delta);
/* one block in place, from pc 0 with a fresh x86 state like compthread does; encode != 0: the compress direction */
int lrzgpu_filter_block(int filter_flag, int delta, uint8_t *data, int64_t n, int encode);
/* sets magic[16] of a 0.13+ image in place (for images whose literal blocks were filtered by the caller) */
int lrzgpu_set_file_filter(uint8_t *lrz, int64_t n, int filter_flag, int delta);

/* ---- scan access hooks: what control->full_tag / next_tag / match_len compute in the reference's
 * non-sliding mode (src/rzip.c:385-393, 405-416, 431-461), over a plain host buffer.  The accelerated forms are
 * lrzgpu_hash_search* (every tag of a chunk in k_tag_scan, match verification inside the resolver); these are the
 * per-position meanings, for a caller that keeps its own loop. */
uint64_t lrzgpu_full_tag(const uint8_t *buf, int64_t p);             /* XOR of hash_index[buf[p .. p + 30]] */
uint64_t lrzgpu_next_tag(const uint8_t *buf, int64_t p, uint64_t t); /* tag at p from the tag at p - 1 */
/* equal bytes forwards from (p0, op) up to `end` plus backwards down to max(0, last_match); 0 if fewer than 31;
 * *rev = the backward part */
int64_t lrzgpu_match_len(const uint8_t *buf, int64_t p0, int64_t op, int64_t end, int64_t last_match, int64_t *rev);
/* The accelerated form of the first two, on its own: every position p in [first, chunk_size - 31] of a chunk in HBM
 * (16-byte aligned, readable 64 bytes past its end) whose tag has all bits of min_mask set -- the candidates of
 * hash_search's loop (src/rzip.c:654-659) as k_tag_scan hands them to the resolver.  *count = how many, *checksum = the
 * sum over them of ((p + 1) * 0x9E3779B97F4A7C15) ^ (tag * 0xC2B2AE3D27D4EB4F) mod 2^64 (order-free: a caller -- the
 * tests -- recomputes it from lrzgpu_full_tag / lrzgpu_next_tag).  reps > 1: the kernels are run that many times and
 * *ms_per_pass is the average of the passes after the first (only_tags: of k_tag_scan alone, without the two list kernels). */
int lrzgpu_tag_candidates_dev(const void *d_chunk, int64_t chunk_size, int64_t first, uint64_t min_mask, int reps,
			      int64_t *count, uint64_t *checksum, double *ms_per_pass, int only_tags, int device);

/* ---- misc ------------------------------------------------------------------------------------------------------
 * lrzgpu_trim() (lrzgpu.h) returns parked buffers and workspaces; the streams the library parks instead of destroying
 * stay open.  lrzgpu_shutdown() = lrzgpu_trim() + those streams closed: once, before exit(). */
void lrzgpu_shutdown(void);
/* CPU seconds the threads of the whole-file pipeline have burnt, by role, since the last call with reset != 0:
 * 0 host encoders (LZMA parser + range coder, or zstd), 1 GPU workers (block copies, finder launches, list copies),
 * 2 scanners (kernel launches, token streams), 3 the whole-input hash, 4 the reader.  (Measurement only.) */
void lrzgpu_profile_cpu(double out[8], int reset);

#ifdef __cplusplus
}
#endif
#endif
