// common.h -- small helpers shared by the host-side translation units.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "lzma_enc.h"

namespace lrzgpu {
int select_device(int device); // 0 or LRZGPU_E_*
int lzma_normalize(LzmaParams &p, int level, unsigned dictSize, int lc, int lp, int pb, int fb);
} // namespace lrzgpu
