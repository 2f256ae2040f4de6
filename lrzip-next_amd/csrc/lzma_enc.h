// lzma_enc.h -- interface of the host half of the LZMA back end (lzma_parser.cpp, lzma_model.h,
// lzma_rangecoder.h).
//
// The GPU match finder (lzma_mf.hip) produces, for every position of a block, the exact
// (len, dist-1) list the reference encoder would receive from its multithreaded BT4 finder
// (reference src/lzma/C/LzFindMt.c:1274-1317).  The host parser consumes those lists and emits the
// raw LZMA stream the reference's encoder (src/lzma/C/LzmaEnc.c) writes, bit for bit.
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

namespace lrzgpu {

// Per-position match lists for one block, in position order:
//   counts[i]  = number of u32 entries of position i (2 per (len, dist-1) pair, lengths increasing)
//   pairs[]    = the entries of position 0, then position 1, ...
// The parser walks positions monotonically and reads the lists in place.
struct MatchLists {
	const uint8_t *counts = nullptr; // n entries
	const uint32_t *pairs = nullptr;
	// tail_flags: the producer also answered, per pair, "do the two bytes after this match and one literal continue
	// at the same distance?" (lzma_mf.hip k_gather): bit 31 of the len word.
	// packed (implies tail_flags): pairs[] holds one u32 per pair: flag << 31 | (len - 2) << 25 | dist-1
	// (counts[] still counts 2 per pair).
	bool packed = false, tail_flags = false;
};

struct LzmaParams {
	int level = 7;
	uint32_t dict_size = 1u << 25;
	int lc = 3, lp = 0, pb = 2;
	int fb = 64;
	bool fast = false; // algo 0 (levels 1-4): GetOptimumFast + the HC5 finder (LzmaEnc.c:95-99)
	uint32_t cut() const { return (16u + ((uint32_t)fb >> 1)) >> (fast ? 1 : 0); } // mc, LzmaEnc.c:99
};

enum : int { LZ_OK = 0, LZ_ERROR_MEM = 2, LZ_ERROR_PARAM = 5, LZ_ERROR_OUTPUT_EOF = 7 };

// Encodes src[0..n) given its match lists. dest_cap bytes available at dest; on success
// *dest_len is the stream size. Returns LZ_ERROR_OUTPUT_EOF when the stream does not fit
// (reference: SeqOutStreamBuf_Write overflow, LzmaEnc.c:2971-2988, 3102-3107).
int lzma_encode_block(const LzmaParams &prm, const uint8_t *src, size_t n, const MatchLists &ml,
		      uint8_t *dest, size_t dest_cap, size_t *dest_len);

// Early start of a block: the parser begins on the lists of the block's first positions while
// the rest of the block is still being scanned, and is handed more in stages.  `early` is valid for positions <
// early_positions (the finder ran on a prefix: lists below its end - fb - 4 equal the whole block's), src[] for bytes
// below the prefix's end.  `rest(ctx, &valid)` is called from the encoding thread whenever the parse comes within reach of
// the current limit (one search window + two maximal match lengths): it blocks until the producer has more and returns
// the lists (same format, positions from 0; the same arrays extended, or new ones) with *valid = the new limit, the
// block's length once everything is there; nullptr withdraws the block (LZ_ERROR_PARAM is returned, nothing of dest is
// meaningful).  Same bytes out as lzma_encode_block() on the whole block's lists.
struct StagedLists {
	MatchLists early;
	size_t early_positions = 0;
	const MatchLists *(*rest)(void *ctx, size_t *valid_positions) = nullptr;
	void *ctx = nullptr;
};
int lzma_encode_block_staged(const LzmaParams &prm, const uint8_t *src, size_t n, const StagedLists &sl,
			     uint8_t *dest, size_t dest_cap, size_t *dest_len);

// 5-byte LZMA properties (reference LzmaEnc_WriteProperties, LzmaEnc.c:3037-3070).
void lzma_write_props(const LzmaParams &prm, uint8_t props[5]);

// Hash mask the reference derives for a block (LzFind.c:347-373, 432-442).
uint32_t lzma_hash_mask(uint32_t dict_size, uint64_t expected_size);
uint32_t lzma_hash_mask5(uint32_t dict_size, uint64_t expected_size); // HC5 finder (levels 1-4)

} // namespace lrzgpu
