// driver.h -- internal interface of the whole-file compress driver (driver.cpp).
#pragma once
#include <sys/types.h>

#include <cstddef>
#include <cstdint>
#include <vector>

#include "../../include/lrzgpu.h"

namespace lrzgpu {

// where the input bytes are: exactly one of host / dev / fd
struct CompressSource {
	const uint8_t *host = nullptr;
	const uint8_t *dev = nullptr;
	// chunk-sharded runs that were handed only their own chunks: dev_chunks[k] = chunk k's bytes on the device
	// (nullptr for chunks of other ranks); then neither host nor dev is set and no whole-input hash can be asked for
	const uint8_t *const *dev_chunks = nullptr;
	int fd = -1;
	int64_t fd_base = 0; // file offset of input byte 0
	int64_t n = 0;
};

// where the container goes: a 21-byte head reserved first and filled in last (the magic), chunks and the
// MD5 appended in file order in between
struct CompressSink {
	virtual int begin(size_t placeholder) = 0;
	virtual int put(const uint8_t *p, size_t n) = 0;
	virtual int finish(const uint8_t *head, size_t n) = 0;
	// n bytes of appended space to be filled in place by the caller (nullptr: not offered, use put())
	virtual uint8_t *append_space(size_t n)
	{
		(void)n;
		return nullptr;
	}
	virtual ~CompressSink() {}
};
// the image in one malloc()ed buffer that is handed to the caller as it is (no copy at the end); grows by
// realloc(), which moves multi-GiB buffers by remapping pages
struct MemorySink : CompressSink {
	uint8_t *p = nullptr;
	size_t len = 0, cap = 0;
	int begin(size_t placeholder) override;
	int put(const uint8_t *q, size_t n) override;
	int finish(const uint8_t *head, size_t n) override;
	uint8_t *append_space(size_t n) override;
	uint8_t *release() // ownership to the caller (free())
	{
		uint8_t *r = p;
		p = nullptr;
		len = cap = 0;
		return r;
	}
	~MemorySink() override;
};
struct FdSink : CompressSink {
	int fd = -1;
	bool with_magic = true; // false: rzip_fd() alone -- chunks + MD5 at the current offset, no magic
	bool seekable = true;
	off_t start = 0;
	std::vector<uint8_t> held;
	int begin(size_t placeholder) override;
	int put(const uint8_t *p, size_t n) override;
	int finish(const uint8_t *head, size_t n) override;
};

// chunk-sharded runs (one process per GPU working on one file): this run compresses the chunks k with
// k % stride == first and hands each finished chunk image to on_chunk instead of a sink
struct ChunkSelect {
	int first = 0, stride = 1;
	const int64_t *victim_in = nullptr; // per chunk index: victim_round to start from, < 0 = predict
	bool with_md5 = false;
	lrzgpu_chunk_fn on_chunk = nullptr;
	void *ctx = nullptr;
};

int run_compress(lrzgpu_control *ctl, const CompressSource &in, CompressSink &out, const ChunkSelect *sel);

} // namespace lrzgpu
