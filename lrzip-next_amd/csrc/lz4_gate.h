// lz4_gate.h -- device lz4 compressibility gate (see lz4_gate.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace lrzgpu {

struct Lz4Job {
	const uint8_t *src; // device pointer
	int src_size;
	int dst_capacity;
	// 0: exact size.  > 0: the caller only needs to know whether the size is below this value; the
	// kernel may stop as soon as that is certain and then returns an upper bound of the size that is
	// itself below stop_below (what is left of the block costs at most rest + rest/255 + 16 bytes).
	int stop_below;
};

// results[j] = liblz4 1.9.3 LZ4_compress_default(src, dst, src_size, dst_capacity) return value.
int lz4_sizes_device(const Lz4Job *d_jobs, int njobs, int *d_results, hipStream_t s);

// reference src/stream.c:2325-2380 decision from a size oracle; size_fn(in_len, d_len) must
// return the LZ4 size of the first in_len bytes of the block.
template <typename SizeFn>
inline int lz4_compresses_decision(long long s_len, int threshold, SizeFn size_fn)
{
	const long long ONE_MB = 1048576, STREAM_BUFSIZE = 10 * ONE_MB;
	long long test_len = s_len;
	int in_len = (int)(test_len < 100 * ONE_MB ? test_len : 100 * ONE_MB);
	int buftest_size = in_len, d_len = in_len + 1;
	double pct = 101;
	while (test_len > 0) {
		int r = size_fn(in_len, d_len);
		if (r > 0) {
			pct = 100 * ((double)r / (double)in_len);
			if (r < in_len * ((double)threshold / 100))
				break;
		}
		test_len -= in_len;
		if (test_len > 0) {
			buftest_size += in_len;
			if (buftest_size < STREAM_BUFSIZE)
				buftest_size <<= 1;
			in_len = (int)(test_len < buftest_size ? test_len : buftest_size);
			d_len = in_len + 1;
		}
	}
	return (int)(pct > threshold ? 0 : pct < 1 ? pct + 1 : pct);
}

} // namespace lrzgpu
