// filters.h -- see filters.cpp.  Filter flags are the container's (magic[16], src/include/lrzip_private.h:389-397).
#pragma once
#include <cstddef>
#include <cstdint>

namespace lrzgpu {

enum FilterFlag {
	FILTER_NONE = 0,
	FILTER_X86 = 1,
	FILTER_ARM = 2,
	FILTER_ARMT = 3,
	FILTER_PPC = 4,
	FILTER_SPARC = 5,
	FILTER_IA64 = 6,
	FILTER_ARM64 = 7,
	FILTER_RISCV = 8,
	FILTER_DELTA = 128
};

bool filter_supported(int flag, int delta);
// One literal block in place, as compthread / ucompthread do it: pc 0, fresh state.  -1: unsupported flag / delta.
int filter_block(int flag, int delta, uint8_t *data, size_t n, bool encode);

} // namespace lrzgpu
