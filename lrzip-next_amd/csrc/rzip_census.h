// rzip_census.h -- exact "no 31-byte window of this chunk occurs twice" (rzip_census.hip)
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace lrzgpu {

struct CensusStats {
	int64_t sample_anchors = 0, sample_equal = 0; // the 1/64 sample: anchors, equal neighbours among their sorted values
	int64_t anchors = 0, equal = 0;               // the full pass (0 / 0 when the sample already answered)
	int64_t cleared = 0;                          // equal values of the full pass that turned out to be chance (no equal windows)
};

// 1: no 31-byte window of d_chunk[0..n) occurs twice (so the rzip scan of it finds no match, whatever its table does);
// 0: some may (or the census had no room): scan as ever; < 0: a HIP error.  d_chunk 16-byte aligned.
int duplicate_census(const uint8_t *d_chunk, int64_t n, int device, hipStream_t s, CensusStats *st = nullptr);

} // namespace lrzgpu
