// lzma_mf.h -- device-side LZMA match finder (see lzma_mf.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

#include "lzma_enc.h"

namespace lrzgpu {

struct MfWorkspace {
	size_t max_n;
	unsigned long long pool_cap; // u32 entries
	uint32_t *key_a, *key_b, *val_a, *val_b, *spos;
	uint32_t *prev2, *prev3;
	uint32_t *seg_start, *seg_len, *seg_start_s, *seg_len_s;
	uint8_t *flags;
	uint32_t *son;
	uint8_t *counts;      // result: entries per position
	uint64_t *tmp_start;
	uint64_t *offsets;    // exclusive scan of counts
	uint32_t *pool_tmp;
	uint32_t *pool_out;   // result: entries in position order
	void *scalars;
	void *prim_tmp;
	size_t prim_bytes;
};

// pool_per_pos: u32 pool entries reserved per input byte (typical text needs ~6-10).
int mf_workspace_create(MfWorkspace **out, size_t max_n, double pool_per_pos);
void mf_workspace_destroy(MfWorkspace *w);
// mode: 0 plain (len, dist-1) couples; 1 the same with the tail flag in bit 31 of len; 2 one u32 per pair
// (flag << 31 | (len - 2) << 25 | dist-1, total_entries/2 words; needs dict <= 32 MiB and fb <= 65) -- see k_gather
// block_n (0 = n): d_src[0..n) is a PREFIX of a block of block_n bytes (the early start of DESIGN.md section 9): the hash
// mask is derived from the block's size, so the buckets -- and with them every list below n - fb - 4 -- are the ones the
// whole block will have (tests/test_oracle_golden.py: test_bt4_lists_are_prefix_computable); the lists of the last
// fb + 4 positions of the prefix are clipped by its end and must not be used.
int mf_run_device(MfWorkspace *w, const uint8_t *d_src, size_t n, uint32_t dict, uint32_t fb, uint32_t cut,
		  hipStream_t s, unsigned long long *total_entries, int mode = 0, bool hc5 = false, size_t block_n = 0);

} // namespace lrzgpu
