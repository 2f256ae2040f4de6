// hashes.h -- the whole-file hashes a .lrz may carry after its last chunk (reference: `hashes[]`, src/main.c:64-79,
// selected by magic[14], computed through libgcrypt in src/rzip.c:943-950, 1195-1219 and checked in
// src/runzip.c:352-440).  Host code, written from the algorithms' specifications (RFC 1321 is md5.h; here
// ISO 3309 CRC-32, RIPEMD-160, FIPS 180-4 SHA-256/384/512, FIPS 202 SHA3-256/512 and SHAKE128/256).
#pragma once
#include <cstddef>
#include <cstdint>
#include <memory>

namespace lrzgpu {

// hash codes of the container (magic[14]); 0 = "CRC" means the chunk CRCs only, nothing is appended
enum HashCode {
	HASH_CRC = 0,
	HASH_MD5 = 1,
	HASH_RIPEMD = 2,
	HASH_SHA256 = 3,
	HASH_SHA384 = 4,
	HASH_SHA512 = 5,
	HASH_SHA3_256 = 6,
	HASH_SHA3_512 = 7,
	HASH_SHAKE128_16 = 8,
	HASH_SHAKE128_32 = 9,
	HASH_SHAKE128_64 = 10,
	HASH_SHAKE256_16 = 11,
	HASH_SHAKE256_32 = 12,
	HASH_SHAKE256_64 = 13,
	HASH_MAX = 13
};

struct Hasher {
	virtual ~Hasher() {}
	virtual void update(const uint8_t *p, size_t n) = 0;
	virtual void finish(uint8_t *out) = 0; // hash_length(code) bytes
};

// bytes the hash of `code` occupies (the reference's table: 4, 16, 20, 32, 48, 64, 32, 64, 16, 32, 64, 16, 32, 64); -1 = unknown code
int hash_length(int code);
const char *hash_label(int code);
// nullptr for an unknown code
std::unique_ptr<Hasher> make_hasher(int code);

} // namespace lrzgpu
