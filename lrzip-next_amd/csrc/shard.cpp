// shard.cpp -- one file across one process per GPU, behind the C ABI: chunk k of the file belongs to rank k mod world.
//
// rzip chunks are independent units of the .lrz format (own header, hash table, CRC, block offsets relative to the
// chunk; src/rzip.c:599-626, src/stream.c:1740-1770).  Every rank runs the whole path for its own chunks
// (run_compress with a ChunkSelect) and ends up with finished chunk images in pinned host memory; what travels is
//   * three integers per chunk (victim_round in / out, image length): one all-reduce per round of the chain check,
//   * the chunk images themselves, to rank 0, in file order: point-to-point send / recv -- the chunk hand-off.
// The one value that crosses a chunk boundary in the reference is insert_hash()'s static victim_round
// (src/rzip.c:308): ranks start their chunks from a prediction (0: the value only moves when one tag value collects
// max_chain_len table entries), rank 0's view of the (in, out) table shows which chunk (if any) started from the
// wrong value, its owner redoes that one chunk, until the chain holds.  Rank 0 lays out magic + chunks + hash.
// The transport is the caller's (three callbacks: RCCL over xGMI in bench.py, gloo in the CPU test, MPI, ...); no
// byte of the data path is computed by it.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <new>
#include <vector>

#include "../../include/lrzgpu.h"
#include "driver.h"
#include "hashes.h"
#include "pools.h"
#include "profile.h"
#include "stream_layer.h"

using namespace lrzgpu;

namespace {

struct Image {
	RawBuf<uint8_t> bytes; // pinned when the pool can give that: the hand-off DMAs straight out of it
	int64_t len = 0, vin = 0, vout = 0;
};
struct Collector {
	std::map<int, std::unique_ptr<Image>> *images;
	int rc = 0;
};
int collect_chunk(void *ctx, int k, int64_t vin, int64_t vout, const uint8_t *img, int64_t len)
{
	Collector *c = (Collector *)ctx;
	try {
		std::unique_ptr<Image> im(new Image());
		im->bytes.alloc((size_t)(len > 0 ? len : 1), true);
		memcpy(im->bytes.data(), img, (size_t)len);
		im->len = len;
		im->vin = vin;
		im->vout = vout;
		(*c->images)[k] = std::move(im);
		return 0;
	} catch (...) {
		c->rc = LRZGPU_E_NOMEM;
		return -1;
	}
}

int protocol(lrzgpu_control *control, int64_t n, const lrzgpu_shard_comm *comm, lrzgpu_shard_compress_fn fn, void *fn_ctx,
	     const uint8_t *digest, uint8_t **out, int64_t *out_len, int64_t *redone_out)
{
	const int rank = comm->rank, world = comm->world;
	// where this rank's seconds go (lrzgpu_profile.shard_s): a flat curve over N must explain itself
	auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
	const double t_start = now();
	double t_own = 0, t_check = 0, t_redo = 0;
	struct Note {
		const double &own, &check, &redo, t_start;
		double t_handoff0 = 0;
		decltype(now) clk;
		~Note()
		{
			ProfileStore &ps = ProfileStore::get();
			std::lock_guard<std::mutex> lk(ps.mu);
			const double t = clk();
			ps.p.shard_s[0] += own;
			ps.p.shard_s[1] += check;
			ps.p.shard_s[2] += redo;
			ps.p.shard_s[3] += t_handoff0 > 0 ? t - t_handoff0 : 0;
			ps.p.shard_s[5] += t - t_start;
		}
	} note{t_own, t_check, t_redo, t_start, 0, now};
	Sizing sz;
	int rc = sizing_for_input(control, n, &sz);
	if (rc)
		return rc;
	std::vector<int64_t> sizes;
	chunk_sizes_for(control, sz, n, &sizes);
	const int n_chunks = (int)sizes.size();
	std::map<int, std::unique_ptr<Image>> images;
	Collector col{&images, 0};
	// A failure of this rank's compressor must not leave the peers blocked in the next collective: it is carried
	// through it (one more word: how many ranks have failed) and every rank leaves together
	int local_rc = 0;
	auto safe_fn = [&](int first, int stride, const int64_t *victim) -> int { // (nothing may skip the next collective)
		try {
			return fn(fn_ctx, first, stride, victim, collect_chunk, &col);
		} catch (const std::bad_alloc &) {
			return LRZGPU_E_NOMEM;
		} catch (...) {
			return LRZGPU_E_INTERNAL;
		}
	};
	if (rank < n_chunks) {
		rc = safe_fn(rank, world, nullptr);
		if (rc || col.rc)
			local_rc = rc ? rc : col.rc;
	}
	t_own = now() - t_start;
	int64_t redone = 0;
	std::vector<int64_t> meta((size_t)n_chunks * 3 + 1);
	for (;;) {
		std::fill(meta.begin(), meta.end(), 0);
		if (!local_rc)
			for (auto &kv : images) {
				meta[(size_t)kv.first * 3 + 0] = kv.second->vin;
				meta[(size_t)kv.first * 3 + 1] = kv.second->vout;
				meta[(size_t)kv.first * 3 + 2] = kv.second->len;
			}
		meta[(size_t)n_chunks * 3] = local_rc ? 1 : 0;
		const double t_c0 = now();
		const int arc = world > 1 ? comm->allreduce_sum_i64(comm->ctx, meta.data(), n_chunks * 3 + 1) : 0;
		(redone ? t_redo : t_check) += now() - t_c0;
		if (arc != 0)
			return local_rc ? local_rc : LRZGPU_E_IO;
		if (meta[(size_t)n_chunks * 3] != 0)
			return local_rc ? local_rc : LRZGPU_E_PEER;
		// the chain of src/rzip.c:308: chunk k must have started from what chunk k - 1 left
		int bad = -1;
		for (int k = 1; k < n_chunks && bad < 0; k++)
			if (meta[(size_t)k * 3] != meta[(size_t)(k - 1) * 3 + 1])
				bad = k;
		if (bad < 0)
			break;
		if (++redone > (int64_t)n_chunks * 4 + 16)
			return LRZGPU_E_INTERNAL; // (cannot happen: every round fixes the first wrong chunk for good; all ranks count alike)
		// only the FIRST wrong chunk is certain to be wrong: its new end value decides about the rest
		if (bad % world == rank) {
			std::vector<int64_t> victim((size_t)n_chunks, -1);
			victim[(size_t)bad] = meta[(size_t)(bad - 1) * 3 + 1];
			images.erase(bad);
			const double t_r0 = now();
			rc = safe_fn(bad, n_chunks > bad + 1 ? n_chunks : bad + 1, victim.data()); // chunk `bad` alone
			t_redo += now() - t_r0;
			if (rc || col.rc)
				local_rc = rc ? rc : col.rc; // reported by the next round's all-reduce
		}
	}
	if (redone_out)
		*redone_out = redone;
	// chunk hand-off to rank 0, in file order
	note.t_handoff0 = now();
	if (rank != 0) {
		for (int k = rank; k < n_chunks; k += world)
			if (comm->send(comm->ctx, 0, images[k]->bytes.data(), images[k]->len) != 0)
				return LRZGPU_E_IO;
		if (out)
			*out = nullptr;
		if (out_len)
			*out_len = 0;
		return 0;
	}
	if (!out || !out_len)
		return LRZGPU_E_PARAM;
	const int hash_len = control->hash_code == 0 ? 0 : hash_length(control->hash_code);
	if (hash_len < 0 || (hash_len && !digest))
		return LRZGPU_E_PARAM;
	size_t total = 21 + (size_t)hash_len;
	for (int k = 0; k < n_chunks; k++)
		total += (size_t)meta[(size_t)k * 3 + 2];
	uint8_t *o = (uint8_t *)malloc(total);
	if (!o)
		return LRZGPU_E_NOMEM;
	write_magic_for(o, control, sz, n, (size_t)n_chunks);
	size_t at = 21;
	for (int k = 0; k < n_chunks; k++) {
		const size_t len = (size_t)meta[(size_t)k * 3 + 2];
		if (k % world == 0)
			memcpy(o + at, images[k]->bytes.data(), len);
		else if (comm->recv(comm->ctx, k % world, o + at, (int64_t)len) != 0) {
			free(o);
			return LRZGPU_E_IO;
		}
		at += len;
	}
	memcpy(o + at, digest, (size_t)hash_len);
	*out = o;
	*out_len = (int64_t)total;
	control->st_size = n;
	control->stream_bufsize = sz.stream_bufsize;
	control->dictSize_used = sz.dict_size;
	control->threads_used = sz.threads;
	return 0;
}

// the library's own per-rank compressor: the chunks k % stride == first of the input through the whole GPU path
struct OwnCompressor {
	lrzgpu_control *control;
	CompressSource src;
	bool first_call = true;
	bool want_hash = false;
};
int own_compress(void *ctx, int first, int stride, const int64_t *victim_in, lrzgpu_chunk_fn on_chunk, void *on_chunk_ctx)
{
	OwnCompressor *c = (OwnCompressor *)ctx;
	ChunkSelect sel;
	sel.first = first;
	sel.stride = stride;
	sel.victim_in = victim_in;
	sel.with_md5 = c->want_hash && c->first_call; // (the hash thread starts with the first call, beside the chunks)
	sel.on_chunk = on_chunk;
	sel.ctx = on_chunk_ctx;
	c->first_call = false;
	MemorySink unused;
	return run_compress(c->control, c->src, unused, &sel);
}

template <typename F> int guard(F &&f)
{
	try {
		return f();
	} catch (const std::bad_alloc &) {
		return LRZGPU_E_NOMEM;
	} catch (...) {
		return LRZGPU_E_INTERNAL;
	}
}
bool comm_ok(const lrzgpu_shard_comm *c)
{
	return c && c->world >= 1 && c->rank >= 0 && c->rank < c->world && (c->world == 1 || (c->allreduce_sum_i64 && c->send && c->recv));
}

} // namespace

extern "C" int lrzgpu_shard_protocol(lrzgpu_control *control, int64_t n, const lrzgpu_shard_comm *comm, lrzgpu_shard_compress_fn fn,
				     void *fn_ctx, const uint8_t *digest, uint8_t **out, int64_t *out_len, int64_t *redone)
{
	if (!control || n < 0 || !comm_ok(comm) || !fn)
		return LRZGPU_E_PARAM;
	return guard([&] { return protocol(control, n, comm, fn, fn_ctx, digest, out, out_len, redone); });
}

static int sharded(lrzgpu_control *control, CompressSource src, int64_t n, const lrzgpu_shard_comm *comm, uint8_t **out, int64_t *out_len,
		   int64_t *redone)
{
	if (!control || n < 0 || !comm_ok(comm))
		return LRZGPU_E_PARAM;
	// the malloc() probe of open_stream_out (control->malloc_probe) sizes blocks by what THIS process may allocate: ranks
	// under different limits would lay the one file out differently -- not in a sharded run
	if (control->malloc_probe)
		return LRZGPU_E_PARAM;
	return guard([&] {
		OwnCompressor oc{control, src};
		oc.src.n = n;
		oc.want_hash = comm->rank == 0 && control->hash_code != 0;
		if (oc.want_hash && oc.src.dev_chunks)
			return (int)LRZGPU_E_PARAM; // rank 0 hashes the whole input: it needs all of it
		int rc = protocol(control, n, comm, own_compress, &oc, control->hash_full, out, out_len, redone);
		return rc;
	});
}

extern "C" int lrzgpu_compress_sharded_dev(lrzgpu_control *control, const void *d_in, int64_t n, const lrzgpu_shard_comm *comm,
					   uint8_t **out, int64_t *out_len, int64_t *redone)
{
	if (!d_in && n)
		return LRZGPU_E_PARAM;
	CompressSource s;
	static const uint8_t empty = 0;
	if (n)
		s.dev = (const uint8_t *)d_in;
	else
		s.host = &empty;
	return sharded(control, s, n, comm, out, out_len, redone);
}

extern "C" int lrzgpu_compress_sharded_chunks_dev(lrzgpu_control *control, const void *const *d_chunks, int64_t n,
						  const lrzgpu_shard_comm *comm, uint8_t **out, int64_t *out_len, int64_t *redone)
{
	if (!d_chunks)
		return LRZGPU_E_PARAM;
	CompressSource s;
	s.dev_chunks = (const uint8_t *const *)d_chunks;
	return sharded(control, s, n, comm, out, out_len, redone);
}

extern "C" int lrzgpu_compress_sharded(lrzgpu_control *control, const uint8_t *in, int64_t n, const lrzgpu_shard_comm *comm, uint8_t **out,
				       int64_t *out_len, int64_t *redone)
{
	if (!in && n)
		return LRZGPU_E_PARAM;
	CompressSource s;
	static const uint8_t empty = 0;
	s.host = in ? in : &empty;
	return sharded(control, s, n, comm, out, out_len, redone);
}
