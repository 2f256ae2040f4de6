// filters_gpu.h -- see filters_gpu.hip: the compress-direction filters over a block resident in HBM.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

namespace lrzgpu {

// device scratch filter_block_device() needs for a block of n bytes (0 for the word / halfword / bundle filters)
size_t filter_scratch_bytes(int flag, size_t n);
// One literal block in place on the device, from pc 0 with fresh state, encode direction; the work is queued on `s`
// (the x86 / RISC-V forms wait once for a count).  0, -1 bad argument, -2 scratch too small, -3 HIP error.
int filter_block_device(int flag, int delta, uint8_t *d_block, size_t n, uint8_t *d_scratch, size_t scratch_bytes, hipStream_t s);

} // namespace lrzgpu
