// stream_layer.h -- host stream layer: sizing, block order and container layout of lrzip-next 0.14.
//
// Mirrors the compress side of reference src/stream.c:
//   prepare_streamout_threads 1090-1118, open_stream_out 1140-1348 (threads / dictionary / limit /
//   stream_bufsize heuristics), write_stream 2198-2216 + flush_buffer 1878-1881 (block boundaries
//   and flush order), compthread 1550-1834 (chunk + block headers), close_stream_out 2253-2282,
// and src/util.c:103-188 (setup_overhead, setup_ram), src/rzip.c:999-1020, 1129-1133 (chunking),
// src/lrzip.c:131-208 (write_magic).
#pragma once
#include <cstdint>
#include <vector>

#include "../../include/lrzgpu.h"

namespace lrzgpu {

constexpr int CTYPE_NONE = 3; // src/include/lrzip_private.h:287-294
constexpr int CTYPE_LZMA = 6;
constexpr int CTYPE_ZSTD = 10;
constexpr int64_t kPage = 4096;

struct Sizing {
	int level = 7, rzip_level = 7;
	bool no_compress = false, lz4_test = true, nobemt = false;
	bool zstd = false;        // --zstd back end (host libzstd); zstd_level 1..22, zstd_strategy 1..9
	int zstd_level = 0, zstd_strategy = 0;
	int threads = 1;          // after prepare_streamout_threads() and open_stream_out() reductions
	uint32_t dict_size = 0;   // possibly reduced
	int64_t overhead = 0;
	int64_t stream_bufsize = 0;
	int64_t max_chunk = 0;    // rzip chunk size
	int64_t max_mmap = 0;     // control->max_mmap: the chunk size of STDIN mode (src/rzip.c:1046, 1075)
	int threshold = 100;
	int64_t malloc_test = 0;  // limit + overhead * threads: what open_stream_out() tries to malloc first (src/stream.c:1292)
	int backoff_steps = 0;    // times a tenth was taken off `limit` because the host refused that (src/stream.c:1293-1303)
};

// Everything the reference derives from (flags, -p, -m, -w, file size) before the first chunk.  st_size is what
// control->st_size holds when open_stream_out() is called first: the file size, or in STDIN mode the size of the
// first chunk; chunk_limit is that call's argument (< 0: derive it, min(max_chunk, st_size)).
int compute_sizing(const lrzgpu_control *c, int64_t st_size, Sizing *out, int64_t chunk_limit = -1);
// The same for a whole run over n input bytes, STDIN mode included: max_chunk is the size of every chunk but the
// last, blocks are sized from the first chunk.
int sizing_for_input(const lrzgpu_control *c, int64_t n, Sizing *out);
// the chunk sizes of an n-byte input in file order (src/rzip.c:1041-1186; STDIN: an input that ends on a chunk
// boundary gets an empty last chunk, mmap_stdin 800-836)
void chunk_sizes_for(const lrzgpu_control *c, const Sizing &s, int64_t n, std::vector<int64_t> *sizes);

inline int chunk_bytes_for(int64_t chunk_size) // src/rzip.c:1129-1133
{
	int bits = 8;
	while (chunk_size >> bits > 0)
		bits++;
	return bits / 8 + (bits % 8 ? 1 : 0);
}

// One flushed stream buffer (a "block"): bytes [off, off+len) of stream `streamno` of its chunk.
struct BlockRef {
	int streamno;
	int64_t off, len;
};

// Replays write_stream()/write_sbstream()/flush_buffer() over a chunk's token stream to obtain the
// blocks in the order the reference hands them to compthreads (which is the file order).
void block_order(const std::vector<uint8_t> &stream0, int chunk_bytes, int64_t stream1_len, int64_t bufsize,
		 std::vector<BlockRef> *blocks);

struct DoneBlock {
	int streamno = 0;
	int c_type = CTYPE_NONE;
	int64_t s_len = 0;
	std::vector<uint8_t> payload; // c_len bytes
};

// Appends one chunk (header, initial stream headers, chained blocks) to `out` (src/stream.c:1716-1821).
void write_chunk(std::vector<uint8_t> *out, int chunk_bytes, bool eof, int64_t chunk_size,
		 const std::vector<DoneBlock> &blocks);

// The same into memory the caller provides (chunk_image_size() bytes).
size_t chunk_image_size(int chunk_bytes, const std::vector<DoneBlock> &blocks);
void write_chunk_raw(uint8_t *out, int chunk_bytes, bool eof, int64_t chunk_size, const std::vector<DoneBlock> &blocks);

void write_magic(uint8_t magic[21], const Sizing &s, int64_t st_size); // src/lrzip.c:131-208
// the same with the per-run selections of the control: hash code, filter byte, and (STDOUT mode) no size when the
// magic went out before the last chunk was known
void write_magic_for(uint8_t magic[21], const lrzgpu_control *c, const Sizing &s, int64_t st_size, size_t n_chunks);
uint8_t filter_magic_byte(int filter_flag, int delta);

} // namespace lrzgpu
