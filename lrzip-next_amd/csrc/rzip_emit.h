// rzip_emit.h -- serialisation of the resolver's match records into the two rzip streams.
#pragma once
#include <cstdint>
#include <vector>

#include "rzip_scan.h"

namespace lrzgpu {

struct EmitResult {
	std::vector<uint8_t> stream0; // tokens, terminator, CRC (src/rzip.c:184-265, 757-760)
	std::vector<CopyRun> runs;    // literal runs: chunk offset -> stream-1 offset
	int64_t stream1_len = 0;
	int64_t matches = 0, match_bytes = 0, literals = 0, literal_bytes = 0;
};

void emit_streams(const std::vector<MatchRec> &recs, int64_t chunk_size, int chunk_bytes, uint32_t crc, EmitResult *out);

} // namespace lrzgpu
