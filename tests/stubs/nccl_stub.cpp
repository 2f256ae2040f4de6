// nccl_stub.cpp -- test infrastructure: an in-process stand-in for the nine RCCL entry points csrc/shard_rccl.cpp uses, so
// that the C transport's N > 1 paths (two communicator ranks: staging pieces, send / receive pairing, the ragged last
// piece, a peer's failure aborting the pending calls of the other) run on a box with ONE GPU: the ranks are threads of
// one process, a "communicator group" is a mailbox in this library.  Calls are synchronous here (they wait for the
// stream they were "enqueued" on, then move the bytes through host memory): ordering and pairing are exercised, overlap
// is not.  Loaded through lrzgpu_rccl_use_library(); never part of the product.
//   NCCL_STUB_FAIL_SEND=<k>: the k-th ncclSend of the process (1-based) fails.
#include <hip/hip_runtime.h>

#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <mutex>
#include <string>
#include <vector>

extern "C" {
typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4 } ncclResult_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef struct StubComm *ncclComm_t;
typedef int ncclDataType_t; // the transport uses ncclUint8 = 1 and ncclInt64 = 4 (nccl.h)
typedef int ncclRedOp_t;
}

namespace {
struct Group {
	std::mutex mu;
	std::condition_variable cv;
	int nranks = 0, arrived = 0;
	bool aborted = false;
	std::map<std::pair<int, int>, std::deque<std::vector<uint8_t>>> mail; // (from, to) -> messages in order
	// all-reduce of int64 sums, one at a time
	std::vector<int64_t> acc, result;
	int ar_in = 0, ar_out = 0;
	uint64_t ar_gen = 0;
};
std::mutex g_mu;
std::map<std::string, Group *> g_groups;
int g_ids = 0, g_sends = 0;
}
struct StubComm {
	Group *g;
	int rank;
};
namespace {
struct Pending {
	bool send;
	void *buf;
	size_t bytes;
	int peer;
	StubComm *c;
	hipStream_t s;
};
thread_local int t_group_depth = 0;
thread_local std::vector<Pending> t_pending;
const std::chrono::seconds kPatience(60);

ncclResult_t do_send(const Pending &p)
{
	{
		std::lock_guard<std::mutex> lk(g_mu);
		const char *f = getenv("NCCL_STUB_FAIL_SEND");
		if (++g_sends == (f ? atoi(f) : -1))
			return ncclSystemError;
	}
	if (hipStreamSynchronize(p.s) != hipSuccess)
		return ncclUnhandledCudaError;
	std::vector<uint8_t> m(p.bytes);
	if (p.bytes && hipMemcpy(m.data(), p.buf, p.bytes, hipMemcpyDeviceToHost) != hipSuccess)
		return ncclUnhandledCudaError;
	Group *g = p.c->g;
	std::lock_guard<std::mutex> lk(g->mu);
	if (g->aborted)
		return ncclSystemError;
	g->mail[{p.c->rank, p.peer}].push_back(std::move(m));
	g->cv.notify_all();
	return ncclSuccess;
}
ncclResult_t do_recv(const Pending &p)
{
	if (hipStreamSynchronize(p.s) != hipSuccess)
		return ncclUnhandledCudaError;
	Group *g = p.c->g;
	std::vector<uint8_t> m;
	{
		std::unique_lock<std::mutex> lk(g->mu);
		auto &q = g->mail[{p.peer, p.c->rank}];
		if (!g->cv.wait_for(lk, kPatience, [&] { return g->aborted || !q.empty(); }) || g->aborted)
			return ncclSystemError;
		m = std::move(q.front());
		q.pop_front();
	}
	if (m.size() != p.bytes) // the two ends cut a message into different pieces
		return ncclInvalidArgument;
	if (p.bytes && hipMemcpy(p.buf, m.data(), p.bytes, hipMemcpyHostToDevice) != hipSuccess)
		return ncclUnhandledCudaError;
	return ncclSuccess;
}
size_t width(ncclDataType_t t) { return t == 4 || t == 5 || t == 8 ? 8 : (t == 2 || t == 3 || t == 7 ? 4 : 1); }
}

extern "C" {
ncclResult_t ncclGetUniqueId(ncclUniqueId *id)
{
	std::lock_guard<std::mutex> lk(g_mu);
	memset(id->internal, 0, sizeof id->internal);
	snprintf(id->internal, sizeof id->internal, "stub-group-%d", ++g_ids);
	return ncclSuccess;
}
ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId id, int rank)
{
	Group *g;
	{
		std::lock_guard<std::mutex> lk(g_mu);
		Group *&slot = g_groups[std::string(id.internal, sizeof id.internal)];
		if (!slot) {
			slot = new Group();
			slot->nranks = nranks;
		}
		g = slot;
	}
	std::unique_lock<std::mutex> lk(g->mu);
	if (g->nranks != nranks || rank < 0 || rank >= nranks)
		return ncclInvalidArgument;
	g->arrived++;
	g->cv.notify_all();
	if (!g->cv.wait_for(lk, kPatience, [&] { return g->aborted || g->arrived >= g->nranks; }) || g->aborted)
		return ncclSystemError;
	*comm = new StubComm{g, rank};
	return ncclSuccess;
}
ncclResult_t ncclCommDestroy(ncclComm_t c)
{
	delete c; // (groups live as long as the process: a test's worth)
	return ncclSuccess;
}
ncclResult_t ncclCommAbort(ncclComm_t c)
{
	{
		std::lock_guard<std::mutex> lk(c->g->mu);
		c->g->aborted = true;
		c->g->cv.notify_all();
	}
	delete c;
	return ncclSuccess;
}
ncclResult_t ncclGroupStart()
{
	t_group_depth++;
	return ncclSuccess;
}
ncclResult_t ncclGroupEnd()
{
	if (--t_group_depth > 0)
		return ncclSuccess;
	ncclResult_t r = ncclSuccess;
	for (const Pending &p : t_pending) // sends first: a rank may send to itself inside a group
		if (p.send && r == ncclSuccess)
			r = do_send(p);
	for (const Pending &p : t_pending)
		if (!p.send && r == ncclSuccess)
			r = do_recv(p);
	t_pending.clear();
	return r;
}
ncclResult_t ncclSend(const void *buf, size_t count, ncclDataType_t type, int peer, ncclComm_t c, hipStream_t s)
{
	const Pending p{true, const_cast<void *>(buf), count * width(type), peer, c, s};
	if (t_group_depth) {
		t_pending.push_back(p);
		return ncclSuccess;
	}
	return do_send(p);
}
ncclResult_t ncclRecv(void *buf, size_t count, ncclDataType_t type, int peer, ncclComm_t c, hipStream_t s)
{
	const Pending p{false, buf, count * width(type), peer, c, s};
	if (t_group_depth) {
		t_pending.push_back(p);
		return ncclSuccess;
	}
	return do_recv(p);
}
ncclResult_t ncclAllReduce(const void *in, void *out, size_t count, ncclDataType_t type, ncclRedOp_t, ncclComm_t c, hipStream_t s)
{
	if (width(type) != 8)
		return ncclInvalidArgument;
	if (hipStreamSynchronize(s) != hipSuccess)
		return ncclUnhandledCudaError;
	std::vector<int64_t> v(count);
	if (count && hipMemcpy(v.data(), in, count * 8, hipMemcpyDeviceToHost) != hipSuccess)
		return ncclUnhandledCudaError;
	Group *g = c->g;
	{
		std::unique_lock<std::mutex> lk(g->mu);
		// the round before must have been read by everybody
		if (!g->cv.wait_for(lk, kPatience, [&] { return g->aborted || g->ar_out == 0; }) || g->aborted)
			return ncclSystemError;
		if (g->ar_in == 0)
			g->acc.assign(count, 0);
		if (g->acc.size() != count)
			return ncclInvalidArgument;
		for (size_t k = 0; k < count; k++)
			g->acc[k] += v[k];
		const uint64_t gen = g->ar_gen;
		if (++g->ar_in == g->nranks) {
			g->result = g->acc;
			g->ar_in = 0;
			g->ar_out = g->nranks;
			g->ar_gen++;
			g->cv.notify_all();
		} else if (!g->cv.wait_for(lk, kPatience, [&] { return g->aborted || g->ar_gen != gen; }) || g->aborted)
			return ncclSystemError;
		v = g->result;
		if (--g->ar_out == 0)
			g->cv.notify_all();
	}
	if (count && hipMemcpy(out, v.data(), count * 8, hipMemcpyHostToDevice) != hipSuccess)
		return ncclUnhandledCudaError;
	return ncclSuccess;
}
}
