// lzma_dec.h -- raw LZMA1 stream decoder (host), the read side of lzma_enc.cpp.
//
// lrzip-next stores every LZMA block as a raw LZMA stream without the 13-byte .lzma header and
// without an end marker; lc/lp/pb/dict come from the file magic and the uncompressed size from the
// block header (reference src/stream.c:744-779 lzma_decompress_buf -> LzmaUncompress,
// src/lzma/C/LzmaDec.c).  Used by the in-library round-trip verifier (unrzip.cpp).
#pragma once
#include <cstddef>
#include <cstdint>

namespace lrzgpu {
// Decodes exactly out_len bytes.  Returns 0, or -1 on corrupt / truncated input.
int lzma_decode_block(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_len, unsigned lc, unsigned lp, unsigned pb);
} // namespace lrzgpu
