// rzip_scan.h -- device-side rzip long-range preprocessor (see rzip_scan.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <functional>
#include <vector>

namespace lrzgpu {

struct MatchRec {
	int64_t p, ofs, len; // match at chunk offset p copies `len` bytes from chunk offset `ofs`
};

// Resolver state, resident in device memory for the lifetime of one chunk scan
// (the automaton state of reference src/rzip.c:586-705 hash_search + struct rzip_state).
struct ScanState {
	// constants of the chunk
	int64_t chunk_size, end;
	int32_t hash_bits;
	uint32_t max_chain_len;
	int64_t hash_limit;
	// automaton
	int64_t p_skip;     // candidates at positions <= p_skip are not examined
	int64_t last_match;
	int64_t cur_p, cur_ofs, cur_len;
	uint64_t tag_mask, min_mask;
	int64_t hash_count, clean_ptr, victim_round;
	// outputs
	int64_t n_records, rec_cap;
	int32_t error; // 1 = record buffer full, 2 = internal, 3 = long forward extent wanted (ext_*), not an error
	int32_t pad;
	// statistics (reference st->stats)
	int64_t inserts, lookups, tag_hits, tag_misses;
	uint64_t sink; // keeps prefetch loads alive
	// long matches: the single resolver wave hands a forward extent that is still equal after
	// LONG_EXTENT bytes to a grid-wide compare kernel; ext_p/ext_op/ext_done is the request,
	// hint_* the answer (equal bytes of chunk[hint_p..] and chunk[hint_op..], up to the chunk end)
	int64_t ext_p, ext_op, ext_done;
	int64_t hint_p, hint_op, hint_len;
	// resolver diagnostics: batches, committed lanes, serial steps, first-stop reasons
	// (complex, real match, conflict, no victim in reach, insert inside swept range)
	int64_t dbg[16]; // [8..15]: shader-clock cycles per phase (refill, simulate, victims, conflict, apply, tail)
};

struct ScanWorkspace {
	int hash_bits;
	int batch_mode;    // 1 = speculative batch resolver, 0 = serial reference path
	void *table;       // 16 B slots
	uint8_t *rank_bytes, *fp_bytes; // one rank / fingerprint byte per slot (see rzip_scan.hip)
	ScanState *state;  // device
	uint64_t *hx;      // device copy of hash_index[256]
	uint32_t *cand_rel;
	uint64_t *cand_tag;
	uint32_t *tile_count;
	uint32_t *tile_base;  // exclusive scan of tile_count (+ total)
	uint32_t *comp_rel;   // the segment's candidates packed in position order
	uint64_t *comp_tag;
	size_t comp_cap;
	size_t seg_cap;    // positions per segment the candidate arrays can hold
	MatchRec *records;
	int64_t rec_cap;
	uint32_t *crc_partial;
	unsigned long long *long_best; // k_long_compare result
	size_t crc_cap;
};

struct ScanResult {
	std::vector<MatchRec> records;
	uint32_t crc;
	ScanState final_state;
};

int scan_workspace_create(ScanWorkspace **out, int rzip_level, int64_t max_chunk);
void scan_workspace_destroy(ScanWorkspace *w);

// Called after every resolver segment (stream already synchronised): `h` is the automaton state,
// `scanned_upto` the last position examined; records [0, h.n_records) in w->records are final.
// A non-zero return aborts the scan with that code.
typedef std::function<int(const ScanState &h, int64_t scanned_upto)> ScanProgressFn;

// Scans d_chunk[0..chunk_size) (device). victim_round in/out.
// census: the caller has no use for the victim_round the chunk ends with (it is the file's last): a chunk in which no
// 31-byte window occurs twice (rzip_census.hip) is not put through the table automaton at all -- no match can come of
// it; the records stay empty and *victim_round is left as it came in.
int scan_chunk_device(ScanWorkspace *w, const uint8_t *d_chunk, int64_t chunk_size, int rzip_level,
		      int64_t *victim_round, ScanResult *res, hipStream_t s, const ScanProgressFn &progress = nullptr, bool census = false);

// K1 alone: candidates of positions [first, end] under min_mask -> d_out[0] = how many, d_out[1] = checksum over
// (position, tag); *ms = one pass of the K1 kernels (only_tags: of k_tag_scan alone) over the range, the average of
// reps - 1 passes after the first
int tag_candidates_device(ScanWorkspace *w, const uint8_t *d_chunk, int64_t first, int64_t end, uint64_t min_mask, int reps,
			  unsigned long long *d_out, double *ms, hipStream_t s, bool only_tags);

// literal gather: dst[dst_off + k] = src[src_off + k] for each run (device pointers)
struct CopyRun {
	int64_t src_off, dst_off, len;
};
// writes destination bytes [dst_lo, dst_hi) only (runs must cover that range, sorted by dst_off)
int gather_runs_device(const uint8_t *d_src, uint8_t *d_dst, const CopyRun *d_runs, int nruns, int64_t dst_lo, int64_t dst_hi, hipStream_t s);

// CRC-32/IEEE of a device buffer
int crc32_device(ScanWorkspace *w, const uint8_t *d_buf, int64_t n, uint32_t *crc, hipStream_t s);

void hash_index_table(uint64_t out[256]);
void rzip_level_params(int level, unsigned *mb_used, unsigned *initial_freq, unsigned *max_chain_len);

} // namespace lrzgpu
