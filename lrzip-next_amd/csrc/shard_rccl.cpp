// shard_rccl.cpp -- the chunk hand-off's transport in C over RCCL (xGMI between the GPUs of a node): the three callbacks
// of lrzgpu_shard_comm (include/lrzgpu.h) for a caller that has no Python in it.
//
//   sum of a few int64 over the ranks   ncclAllReduce(ncclInt64, ncclSum) on a device copy of the words
//   send bytes to a rank                pieces of kPiece bytes: pinned host -> staging buffer (copy stream) -> ncclSend
//                                       (comm stream); two staging buffers, so the copy of piece i + 1 runs beside the
//                                       send of piece i
//   receive bytes from a rank           ncclRecv into a staging buffer -> host; the receive of piece i + 1 is queued
//                                       before piece i is copied out
// Nothing of the data path is computed here (shard.cpp says what travels: three integers per chunk per round of the
// chain check, and the finished chunk images to rank 0).  RCCL is taken from the process at run time (dlopen of
// librccl.so.1 -- the one torch has already loaded when the caller is bench.py): liblrzgpu.so itself links no
// communication library and loads on a box without one; lrzgpu_rccl_* then return LRZGPU_E_NODEVICE.
// The unique id (128 bytes, rank 0: lrzgpu_rccl_unique_id) reaches the other ranks by the caller's own bootstrap
// (MPI_Bcast, a file, a TCP store ...), as with ncclGetUniqueId / ncclCommInitRank themselves.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstring>
#include <ctime>
#include <mutex>
#include <new>
#include <string>

#include "../../include/lrzgpu.h"
#include "common.h"

using namespace lrzgpu;

namespace {

struct Rccl {
	void *h = nullptr;
	ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
	ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
	ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
	ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
	ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
	ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
	ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
	ncclResult_t (*GroupStart)() = nullptr;
	ncclResult_t (*GroupEnd)() = nullptr;
	bool ok = false;
	std::string loaded_from; // the path asked for with lrzgpu_rccl_use_library(), or empty: the process's librccl
};
// the library to take the entry points from instead of librccl (lrzgpu_rccl_use_library); read once, under its mutex,
// by the initialisation below
std::mutex &override_mu()
{
	static std::mutex m;
	return m;
}
std::string &library_override()
{
	static std::string path;
	return path;
}
Rccl &rccl()
{
	static Rccl r;
	static std::once_flag once;
	std::call_once(once, [] {
		{
			std::lock_guard<std::mutex> lk(override_mu());
			r.loaded_from = library_override();
		}
		if (!r.loaded_from.empty())
			r.h = dlopen(r.loaded_from.c_str(), RTLD_NOW | RTLD_LOCAL);
		else
			for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
				r.h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
				if (r.h)
					break;
			}
		if (!r.h)
			return;
		auto sym = [&](const char *s) { return dlsym(r.h, s); };
		r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
		r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
		r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
		r.CommAbort = (decltype(r.CommAbort))sym("ncclCommAbort");
		r.AllReduce = (decltype(r.AllReduce))sym("ncclAllReduce");
		r.Send = (decltype(r.Send))sym("ncclSend");
		r.Recv = (decltype(r.Recv))sym("ncclRecv");
		r.GroupStart = (decltype(r.GroupStart))sym("ncclGroupStart");
		r.GroupEnd = (decltype(r.GroupEnd))sym("ncclGroupEnd");
		r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.CommAbort && r.AllReduce && r.Send && r.Recv && r.GroupStart &&
		       r.GroupEnd;
	});
	return r;
}

constexpr size_t kPiece = (size_t)32 << 20; // both ends cut a message into the same pieces

struct Transport {
	ncclComm_t comm = nullptr;
	int device = 0, rank = 0, world = 1;
	hipStream_t comm_stream = nullptr, copy_stream = nullptr;
	uint8_t *stage[2] = {nullptr, nullptr};
	hipEvent_t in_stage[2] = {nullptr, nullptr};  // the piece is in the staging buffer (copied in / received)
	hipEvent_t out_stage[2] = {nullptr, nullptr}; // the piece has left it (sent / copied out)
	hipEvent_t reduced = nullptr;                 // the all-reduce's result is on the device
	int64_t *d_words = nullptr;
	size_t d_words_cap = 0;
	bool broken = false;

	~Transport()
	{
		(void)hipSetDevice(device);
		if (comm) {
			if (broken)
				rccl().CommAbort(comm); // (a peer may be gone: destroy would wait for it)
			else
				rccl().CommDestroy(comm);
		}
		for (int b = 0; b < 2; b++) {
			if (stage[b])
				(void)hipFree(stage[b]);
			if (in_stage[b])
				(void)hipEventDestroy(in_stage[b]);
			if (out_stage[b])
				(void)hipEventDestroy(out_stage[b]);
		}
		if (reduced)
			(void)hipEventDestroy(reduced);
		if (d_words)
			(void)hipFree(d_words);
		if (comm_stream)
			(void)hipStreamDestroy(comm_stream);
		if (copy_stream)
			(void)hipStreamDestroy(copy_stream);
	}
	int fail()
	{
		broken = true;
		if (comm) { // the peers' pending calls on this communicator fail instead of waiting for ever
			rccl().CommAbort(comm);
			comm = nullptr;
		}
		return -1;
	}
};

// A send / receive whose peer never shows up must not hang the job for ever: waits on the communication streams give up
// after this long, the communicator is aborted (which fails the peer's pending calls too) and the callback reports an
// error -- the protocol then ends with LRZGPU_E_IO on this rank.
constexpr double kPeerTimeoutSeconds = 600.0;
// ... and the all-reduce is where a rank that is through waits for the slowest one (a whole chunk compressed, or redone
// after a wrong victim_round guess): six times that
constexpr double kReduceTimeoutSeconds = 3600.0;
hipError_t wait_event_bounded(hipEvent_t ev, double patience = kPeerTimeoutSeconds)
{
	timespec t0, t;
	clock_gettime(CLOCK_MONOTONIC, &t0);
	long ns = 30000;
	for (;;) {
		const hipError_t q = hipEventQuery(ev);
		if (q != hipErrorNotReady)
			return q;
		clock_gettime(CLOCK_MONOTONIC, &t);
		if ((double)(t.tv_sec - t0.tv_sec) + 1e-9 * (double)(t.tv_nsec - t0.tv_nsec) > patience)
			return hipErrorNotReady;
		timespec ts{0, ns};
		nanosleep(&ts, nullptr);
		if (ns < 2000000)
			ns += ns / 4;
	}
}

#define HIPOK(x)                   \
	do {                       \
		if ((x) != hipSuccess) \
			return t->fail();  \
	} while (0)
#define NCCLOK(x)                   \
	do {                        \
		if ((x) != ncclSuccess) \
			return t->fail();   \
	} while (0)

int cb_allreduce(void *ctx, int64_t *vals, int count)
{
	Transport *t = (Transport *)ctx;
	if (t->broken || count < 0)
		return -1;
	if (count == 0)
		return 0;
	HIPOK(hipSetDevice(t->device));
	if ((size_t)count > t->d_words_cap) {
		if (t->d_words)
			(void)hipFree(t->d_words);
		t->d_words = nullptr;
		t->d_words_cap = 0;
		HIPOK(hipMalloc((void **)&t->d_words, (size_t)count * 8));
		t->d_words_cap = (size_t)count;
	}
	HIPOK(hipMemcpyAsync(t->d_words, vals, (size_t)count * 8, hipMemcpyHostToDevice, t->comm_stream));
	NCCLOK(rccl().AllReduce(t->d_words, t->d_words, (size_t)count, ncclInt64, ncclSum, t->comm, t->comm_stream));
	HIPOK(hipEventRecord(t->reduced, t->comm_stream));
	HIPOK(wait_event_bounded(t->reduced, kReduceTimeoutSeconds)); // (sleeps; gives up when a rank never joins)
	HIPOK(hipMemcpy(vals, t->d_words, (size_t)count * 8, hipMemcpyDeviceToHost));
	return 0;
}

// pieces [o, o + k) of a message of n bytes
inline size_t pieces_of(int64_t n)
{
	return (size_t)((n + (int64_t)kPiece - 1) / (int64_t)kPiece);
}

int send_to(Transport *t, int dst, const void *buf, int64_t n, bool loop, void *loop_dst);

int cb_send(void *ctx, int dst, const void *buf, int64_t n)
{
	Transport *t = (Transport *)ctx;
	if (t->broken || n < 0 || dst < 0 || dst >= t->world || dst == t->rank)
		return -1;
	return send_to(t, dst, buf, n, false, nullptr);
}

// host -> staging (copy stream) -> peer (comm stream); with loop_dst: the peer is this rank itself and every piece is
// one group of a send and a receive into the other half of the staging pair's twin (the loop-back self test)
int send_to(Transport *t, int dst, const void *buf, int64_t n, bool loop, void *loop_dst)
{
	HIPOK(hipSetDevice(t->device));
	const uint8_t *src = (const uint8_t *)buf;
	const size_t np = pieces_of(n);
	for (size_t i = 0; i < np; i++) {
		const int b = (int)(i & 1);
		const size_t o = i * kPiece, k = (size_t)n - o < kPiece ? (size_t)n - o : kPiece;
		if (!loop) {
			// the piece sent from this buffer two pieces ago must have left it: waited for here, on the host and with a
			// bound (a copy from pageable memory would make the runtime wait for it instead, for as long as it takes)
			if (i >= 2)
				HIPOK(wait_event_bounded(t->out_stage[b]));
			HIPOK(hipStreamWaitEvent(t->copy_stream, t->out_stage[b], 0)); // (never recorded yet: no wait)
			HIPOK(hipMemcpyAsync(t->stage[b], src + o, k, hipMemcpyHostToDevice, t->copy_stream));
			HIPOK(hipEventRecord(t->in_stage[b], t->copy_stream));
			HIPOK(hipStreamWaitEvent(t->comm_stream, t->in_stage[b], 0));
			NCCLOK(rccl().Send(t->stage[b], k, ncclUint8, dst, t->comm, t->comm_stream));
			HIPOK(hipEventRecord(t->out_stage[b], t->comm_stream));
		} else { // one piece at a time through both buffers: stage[0] -> self -> stage[1]
			HIPOK(hipMemcpyAsync(t->stage[0], src + o, k, hipMemcpyHostToDevice, t->comm_stream));
			NCCLOK(rccl().GroupStart());
			ncclResult_t r1 = rccl().Send(t->stage[0], k, ncclUint8, t->rank, t->comm, t->comm_stream);
			ncclResult_t r2 = rccl().Recv(t->stage[1], k, ncclUint8, t->rank, t->comm, t->comm_stream);
			NCCLOK(rccl().GroupEnd());
			NCCLOK(r1);
			NCCLOK(r2);
			HIPOK(d2h_pageable((uint8_t *)loop_dst + o, t->stage[1], k, t->comm_stream));
		}
	}
	if (!loop && np) // (the last piece's send: out_stage of its buffer; the one before it was waited for by the copy stream)
		HIPOK(wait_event_bounded(t->out_stage[(np - 1) & 1]));
	return 0;
}

int cb_recv(void *ctx, int src, void *buf, int64_t n)
{
	Transport *t = (Transport *)ctx;
	if (t->broken || n < 0 || src < 0 || src >= t->world || src == t->rank)
		return -1;
	HIPOK(hipSetDevice(t->device));
	uint8_t *dst = (uint8_t *)buf;
	const size_t np = pieces_of(n);
	auto queue_recv = [&](size_t i) -> int {
		const int b = (int)(i & 1);
		const size_t o = i * kPiece, k = (size_t)n - o < kPiece ? (size_t)n - o : kPiece;
		HIPOK(hipStreamWaitEvent(t->comm_stream, t->out_stage[b], 0));
		NCCLOK(rccl().Recv(t->stage[b], k, ncclUint8, src, t->comm, t->comm_stream));
		HIPOK(hipEventRecord(t->in_stage[b], t->comm_stream));
		return 0;
	};
	if (np && queue_recv(0))
		return -1;
	for (size_t i = 0; i < np; i++) {
		const int b = (int)(i & 1);
		const size_t o = i * kPiece, k = (size_t)n - o < kPiece ? (size_t)n - o : kPiece;
		if (i + 1 < np && queue_recv(i + 1)) // (its buffer was copied out in the previous iteration)
			return -1;
		// the destination is the caller's (rank 0 lays the .lrz out in malloc'd memory): sleep until the piece is
		// there, then copy it -- a copy into pageable memory makes the thread spin for what is queued before it
		HIPOK(wait_event_bounded(t->in_stage[b])); // (sleeps until the piece is there; gives up when the peer never sends)
		HIPOK(hipMemcpyAsync(dst + o, t->stage[b], k, hipMemcpyDeviceToHost, t->copy_stream));
		HIPOK(hipEventRecord(t->out_stage[b], t->copy_stream));
		HIPOK(stream_wait(t->copy_stream));
	}
	return 0;
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

} // namespace

// The nccl* entry points from another shared object than librccl.so.1 (a build under another name; tests: an in-process
// stand-in that lets two ranks be two threads of one process on one GPU).  Before the first lrzgpu_rccl_* call only.
extern "C" int lrzgpu_rccl_use_library(const char *path)
{
	if (!path || !*path)
		return LRZGPU_E_PARAM;
	{
		std::lock_guard<std::mutex> lk(override_mu());
		library_override() = path;
	}
	// (an initialisation that has already happened -- from librccl, or from another path -- is not undone: the call
	// then fails instead of reporting a library that was never loaded)
	const Rccl &r = rccl();
	return r.ok && r.h && r.loaded_from == path ? 0 : LRZGPU_E_NODEVICE;
}

extern "C" int lrzgpu_rccl_available(void)
{
	return rccl().ok ? 1 : 0;
}

extern "C" int lrzgpu_rccl_unique_id(uint8_t id[LRZGPU_RCCL_ID_BYTES])
{
	static_assert(LRZGPU_RCCL_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "id size");
	if (!id)
		return LRZGPU_E_PARAM;
	if (!rccl().ok)
		return LRZGPU_E_NODEVICE;
	ncclUniqueId u;
	if (rccl().GetUniqueId(&u) != ncclSuccess)
		return LRZGPU_E_IO;
	memcpy(id, u.internal, NCCL_UNIQUE_ID_BYTES);
	return 0;
}

extern "C" int lrzgpu_rccl_comm_create(const uint8_t id[LRZGPU_RCCL_ID_BYTES], int rank, int world, int device, lrzgpu_shard_comm *out)
{
	if (!id || !out || world < 1 || rank < 0 || rank >= world)
		return LRZGPU_E_PARAM;
	if (!rccl().ok)
		return LRZGPU_E_NODEVICE;
	return guard([&]() -> int {
		int rc = select_device(device);
		if (rc)
			return rc;
		Transport *t = new Transport();
		t->device = device;
		t->rank = rank;
		t->world = world;
		auto bail = [&](int code) {
			delete t;
			return code;
		};
		if (hipStreamCreateWithFlags(&t->comm_stream, hipStreamNonBlocking) != hipSuccess ||
		    hipStreamCreateWithFlags(&t->copy_stream, hipStreamNonBlocking) != hipSuccess)
			return bail(LRZGPU_E_HIP);
		if (hipEventCreateWithFlags(&t->reduced, hipEventDisableTiming) != hipSuccess)
			return bail(LRZGPU_E_HIP);
		for (int b = 0; b < 2; b++)
			if (hipMalloc((void **)&t->stage[b], kPiece) != hipSuccess ||
			    hipEventCreateWithFlags(&t->in_stage[b], hipEventDisableTiming) != hipSuccess ||
			    hipEventCreateWithFlags(&t->out_stage[b], hipEventDisableTiming) != hipSuccess)
				return bail(LRZGPU_E_NOMEM);
		ncclUniqueId u;
		memcpy(u.internal, id, NCCL_UNIQUE_ID_BYTES);
		if (rccl().CommInitRank(&t->comm, world, u, rank) != ncclSuccess) {
			t->comm = nullptr;
			return bail(LRZGPU_E_IO);
		}
		out->ctx = t;
		out->rank = rank;
		out->world = world;
		out->allreduce_sum_i64 = cb_allreduce;
		out->send = cb_send;
		out->recv = cb_recv;
		return 0;
	});
}

extern "C" int lrzgpu_rccl_comm_destroy(lrzgpu_shard_comm *comm)
{
	if (!comm || !comm->ctx || comm->allreduce_sum_i64 != cb_allreduce)
		return LRZGPU_E_PARAM;
	delete (Transport *)comm->ctx;
	memset(comm, 0, sizeof *comm);
	return 0;
}

/* Self test of the send / receive path on one rank (a communicator of any size): n bytes src -> staging -> ncclSend to
 * this rank itself, grouped with the matching ncclRecv -> staging -> dst, piece by piece.  What a world of one can
 * exercise of the hand-off; the protocol itself never sends to its own rank. */
extern "C" int lrzgpu_rccl_loopback(lrzgpu_shard_comm *comm, const void *src, void *dst, int64_t n)
{
	if (!comm || !comm->ctx || comm->allreduce_sum_i64 != cb_allreduce || n < 0 || (n && (!src || !dst)))
		return LRZGPU_E_PARAM;
	Transport *t = (Transport *)comm->ctx;
	if (t->broken)
		return LRZGPU_E_IO;
	return guard([&] { return send_to(t, t->rank, src, n, true, dst) ? (int)LRZGPU_E_IO : 0; });
}
