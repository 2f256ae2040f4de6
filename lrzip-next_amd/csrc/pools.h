// pools.h -- process-wide caches of the big buffers the compress path needs again and again.
//
// A compressor that handles one file after another (or the reference's compthread calling the
// per-block entry points 257 times per chunk) must not pay hipMalloc/hipFree, pinning, mmap and
// page faults per call: hipFree synchronises the whole device, a 1.8 GB host buffer costs a page fault
// per 4 KiB.  Everything here is take/give; memory is returned to the system by lrzgpu_trim() or at
// process exit.
#pragma once
#include <hip/hip_runtime.h>
#include <unistd.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <new>
#include <vector>

#include "lzma_mf.h"
#include "rzip_scan.h"

namespace lrzgpu {

// ---- host buffers (pageable or pinned) ---------------------------------------------------------
struct HostPool {
	struct Entry {
		size_t cap;
		void *p;
		bool pinned;
	};
	std::mutex mu;
	std::vector<Entry> idle;
	size_t idle_bytes = 0;
	static HostPool &get()
	{
		static HostPool p;
		return p;
	}
	static size_t idle_limit() // keep at most 1/8 of physical memory (and at most 64 GiB) parked
	{
		static const size_t lim = [] {
			const long pages = sysconf(_SC_PHYS_PAGES), psz = sysconf(_SC_PAGE_SIZE);
			size_t phys = pages > 0 && psz > 0 ? (size_t)pages * (size_t)psz : (size_t)64 << 30;
			size_t l = phys / 8;
			return l > ((size_t)64 << 30) ? (size_t)64 << 30 : l;
		}();
		return lim;
	}
	// pinned: page-locked memory the GPU can DMA into directly (falls back to pageable: *got_pinned)
	void *take(size_t bytes, size_t *cap, bool pinned, bool *got_pinned)
	{
		{
			std::lock_guard<std::mutex> lk(mu);
			size_t best = idle.size();
			for (size_t i = 0; i < idle.size(); i++)
				if (idle[i].pinned == pinned && idle[i].cap >= bytes && (best == idle.size() || idle[i].cap < idle[best].cap))
					best = i;
			if (best != idle.size() && idle[best].cap <= 2 * bytes + ((size_t)64 << 20)) {
				void *p = idle[best].p;
				*cap = idle[best].cap;
				*got_pinned = idle[best].pinned;
				idle_bytes -= idle[best].cap;
				idle[best] = idle.back();
				idle.pop_back();
				return p;
			}
		}
		const size_t round = (size_t)8 << 20;
		*cap = (bytes + round - 1) / round * round;
		if (*cap == 0)
			*cap = round;
		if (pinned) {
			void *p = nullptr;
			if (hipHostMalloc(&p, *cap, hipHostMallocDefault) == hipSuccess && p) {
				*got_pinned = true;
				return p;
			}
			(void)hipGetLastError();
		}
		*got_pinned = false;
		return malloc(*cap);
	}
	void give(void *p, size_t cap, bool pinned)
	{
		if (!p)
			return;
		{
			std::lock_guard<std::mutex> lk(mu);
			if (idle.size() < 256 && idle_bytes + cap <= idle_limit()) {
				idle.push_back(Entry{cap, p, pinned});
				idle_bytes += cap;
				return;
			}
		}
		release(p, pinned);
	}
	static void release(void *p, bool pinned)
	{
		if (pinned)
			(void)hipHostFree(p);
		else
			free(p);
	}
	void trim()
	{
		std::vector<Entry> old;
		{
			std::lock_guard<std::mutex> lk(mu);
			old.swap(idle);
			idle_bytes = 0;
		}
		for (auto &e : old)
			release(e.p, e.pinned);
	}
	~HostPool()
	{
		for (auto &e : idle)
			if (!e.pinned) // the HIP runtime may already be gone at exit: pinned pieces are left to it
				free(e.p);
	}
};

template <typename T> struct RawBuf { // uninitialised host buffer (std::vector would zero-fill), recycled
	T *p = nullptr;
	size_t n = 0, cap = 0;
	bool pinned = false;
	RawBuf() = default;
	RawBuf(const RawBuf &) = delete;
	RawBuf &operator=(const RawBuf &) = delete;
	void alloc(size_t k, bool want_pinned = false)
	{
		release();
		p = (T *)HostPool::get().take((k ? k : 1) * sizeof(T), &cap, want_pinned, &pinned);
		if (!p)
			throw std::bad_alloc();
		n = k;
	}
	void release()
	{
		if (p)
			HostPool::get().give(p, cap, pinned);
		p = nullptr;
		n = cap = 0;
		pinned = false;
	}
	~RawBuf() { release(); }
	T *data() { return p; }
	const T *data() const { return p; }
};

// ---- device memory: never let the runtime run out ---------------------------------------------------
// When an allocation fails, the HSA runtime of this ROCm build trims its own caches and crashes doing so
// (rocr::AMD::GpuAgent::Trim -> AqlQueue::AsyncReclaimMainScratch through a queue that no longer exists:
// seen as SIGSEGV inside hipMalloc in a process that had created and destroyed CU-masked streams).  So the pools
// do not let it get there: every large allocation of the library is made under one mutex after asking the
// driver how much is free, parked buffers are given back first when the device is short, and what is parked is
// bounded.
inline void pools_release_device(int device); // everything parked in DevicePool and WorkspacePool, defined below
struct DeviceBudget {
	static std::mutex &alloc_mu()
	{
		static std::mutex m;
		return m;
	}
	static size_t margin() { return (size_t)6 << 30; }
	static size_t free_now()
	{
		size_t fr = 0, tot = 0;
		if (hipMemGetInfo(&fr, &tot) != hipSuccess) {
			(void)hipGetLastError();
			return ~(size_t)0;
		}
		return fr;
	}
	static size_t idle_limit() // parked device memory above this is freed when it is given back
	{
		static const size_t lim = [] {
			size_t fr = 0, tot = 0;
			if (hipMemGetInfo(&fr, &tot) != hipSuccess || tot == 0) {
				(void)hipGetLastError();
				return (size_t)128 << 30;
			}
			return tot / 10 * 7;
		}();
		return lim;
	}
	// call with alloc_mu held, before hipMalloc()ing `bytes`: false = not even after releasing what is parked
	static bool make_room(size_t bytes, int device)
	{
		if (bytes < ((size_t)256 << 20))
			return true; // (small requests are not worth a driver query each; with no stream ever destroyed the
				     // runtime's own out-of-memory path is survivable again)
		const size_t f0 = free_now();
		if (f0 >= bytes + margin())
			return true;
		pools_release_device(device);
		const size_t f1 = free_now();
		if (getenv("LRZGPU_TRACE"))
			fprintf(stderr, "lrzgpu pools: %zu MiB wanted, %zu MiB free: parked buffers released, now %zu MiB free\n", bytes >> 20, f0 >> 20, f1 >> 20);
		return f1 >= bytes + ((size_t)1 << 30);
	}
};

// ---- streams: created once, parked, never destroyed ---------------------------------------------------
// (a destroyed stream's hardware queue is exactly what the runtime trips over later, see DeviceBudget; a
// compressor that handles file after file needs the same handful of streams again anyway)
struct StreamPool {
	struct Entry {
		hipStream_t s;
		int device, kind; // kind: 0 plain non-blocking, 1 high priority, 16 + k the k-th CU-masked set
	};
	std::mutex mu;
	std::vector<Entry> idle, known;
	static StreamPool &get()
	{
		static StreamPool p;
		return p;
	}
	hipStream_t take(int device, int kind)
	{
		std::lock_guard<std::mutex> lk(mu);
		for (size_t i = 0; i < idle.size(); i++)
			if (idle[i].device == device && idle[i].kind == kind) {
				hipStream_t s = idle[i].s;
				idle[i] = idle.back();
				idle.pop_back();
				return s;
			}
		return nullptr;
	}
	void created(hipStream_t s, int device, int kind)
	{
		std::lock_guard<std::mutex> lk(mu);
		known.push_back(Entry{s, device, kind});
	}
	void give(hipStream_t s) // in place of hipStreamDestroy(): the stream must be idle
	{
		if (!s)
			return;
		std::lock_guard<std::mutex> lk(mu);
		for (auto &e : known)
			if (e.s == s) {
				idle.push_back(e);
				return;
			}
		// not one of ours: leave it alone (and alive)
	}
	void destroy_idle() // lrzgpu_trim(): for a caller about to exit (profilers want to see every queue closed)
	{
		std::vector<Entry> old;
		{
			std::lock_guard<std::mutex> lk(mu);
			old.swap(idle);
			for (auto &e : old)
				for (size_t i = 0; i < known.size(); i++)
					if (known[i].s == e.s) {
						known[i] = known.back();
						known.pop_back();
						break;
					}
		}
		for (auto &e : old)
			(void)hipStreamDestroy(e.s);
	}
};
inline int current_device_or0()
{
	int d = 0;
	if (hipGetDevice(&d) != hipSuccess) {
		(void)hipGetLastError();
		d = 0;
	}
	return d;
}
// a plain non-blocking stream on `device` (the current one when < 0) from the pool; give it back with
// StreamPool::get().give(), never hipStreamDestroy()
inline hipStream_t pooled_stream(int device = -1)
{
	const int dev = device >= 0 ? device : current_device_or0();
	hipStream_t s = StreamPool::get().take(dev, 0);
	if (s)
		return s;
	if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) {
		(void)hipGetLastError();
		return nullptr;
	}
	StreamPool::get().created(s, dev, 0);
	return s;
}

// ---- device buffers ------------------------------------------------------------------------------
struct DevicePool {
	struct Entry {
		size_t cap;
		void *p;
		int device;
	};
	std::mutex mu;
	std::vector<Entry> idle; // oldest first
	size_t idle_bytes = 0;
	static DevicePool &get()
	{
		static DevicePool p;
		return p;
	}
	void *take(size_t bytes, size_t *cap, int device)
	{
		{
			std::lock_guard<std::mutex> lk(mu);
			size_t best = idle.size();
			for (size_t i = 0; i < idle.size(); i++)
				if (idle[i].device == device && idle[i].cap >= bytes && (best == idle.size() || idle[i].cap < idle[best].cap))
					best = i;
			if (best != idle.size() && idle[best].cap <= bytes + bytes / 4 + ((size_t)16 << 20)) {
				void *p = idle[best].p;
				*cap = idle[best].cap;
				idle_bytes -= idle[best].cap;
				idle.erase(idle.begin() + (long)best);
				return p;
			}
		}
		const size_t round = (size_t)2 << 20;
		*cap = (bytes + round - 1) / round * round;
		if (*cap == 0)
			*cap = round;
		void *p = nullptr;
		std::lock_guard<std::mutex> al(DeviceBudget::alloc_mu());
		if (!DeviceBudget::make_room(*cap, device))
			return nullptr;
		if (hipMalloc(&p, *cap) != hipSuccess) {
			(void)hipGetLastError();
			return nullptr;
		}
		return p;
	}
	void give(void *p, size_t cap, int device)
	{
		if (!p)
			return;
		std::vector<Entry> out;
		{
			std::lock_guard<std::mutex> lk(mu);
			idle.push_back(Entry{cap, p, device});
			idle_bytes += cap;
			while (idle_bytes > DeviceBudget::idle_limit() / 2 && idle.size() > 1) {
				out.push_back(idle.front());
				idle_bytes -= idle.front().cap;
				idle.erase(idle.begin());
			}
		}
		for (auto &e : out)
			(void)hipFree(e.p);
	}
	void trim(int device = -1)
	{
		std::vector<Entry> old, keep;
		{
			std::lock_guard<std::mutex> lk(mu);
			for (auto &e : idle)
				(device < 0 || e.device == device ? old : keep).push_back(e);
			idle.swap(keep);
			idle_bytes = 0;
			for (auto &e : idle)
				idle_bytes += e.cap;
		}
		for (auto &e : old)
			(void)hipFree(e.p);
	}
};

struct DevBuf {
	uint8_t *p = nullptr;
	size_t cap = 0;
	int device = 0;
	DevBuf() = default;
	DevBuf(const DevBuf &) = delete;
	DevBuf &operator=(const DevBuf &) = delete;
	bool alloc(size_t bytes, int dev)
	{
		release();
		device = dev;
		p = (uint8_t *)DevicePool::get().take(bytes, &cap, dev);
		return p != nullptr;
	}
	void release()
	{
		if (p)
			DevicePool::get().give(p, cap, device);
		p = nullptr;
		cap = 0;
	}
	~DevBuf() { release(); }
};

// ---- whole workspaces (dozens of allocations each) ---------------------------------------------
struct WorkspacePool {
	struct MfEntry {
		MfWorkspace *w;
		int device;
		double per_pos;
	};
	struct ScanEntry {
		ScanWorkspace *w;
		int device, level;
		int64_t max_chunk;
	};
	std::mutex mu;
	std::vector<MfEntry> mf; // oldest first
	std::vector<ScanEntry> scan;
	size_t idle_bytes = 0;
	static size_t mf_bytes(size_t max_n, double per_pos) // what mf_workspace_create allocates, roughly
	{
		return max_n * 110 + (size_t)((double)max_n * per_pos) * 8 + ((size_t)64 << 20);
	}
	static size_t scan_bytes(const ScanWorkspace *w)
	{
		return ((size_t)18 << w->hash_bits) + w->seg_cap * 12 + w->comp_cap * 12 + (size_t)w->rec_cap * sizeof(MatchRec) + ((size_t)16 << 20);
	}
	// parked workspaces above the bound are destroyed, oldest first (call with mu held; destroy outside)
	void evict_locked(std::vector<MfEntry> &m_out, std::vector<ScanEntry> &s_out)
	{
		while (idle_bytes > DeviceBudget::idle_limit() && (mf.size() + scan.size()) > 1) {
			if (!mf.empty() && (scan.empty() || mf_bytes(mf.front().w->max_n, mf.front().per_pos) >= scan_bytes(scan.front().w))) {
				idle_bytes -= mf_bytes(mf.front().w->max_n, mf.front().per_pos);
				m_out.push_back(mf.front());
				mf.erase(mf.begin());
			} else {
				idle_bytes -= scan_bytes(scan.front().w);
				s_out.push_back(scan.front());
				scan.erase(scan.begin());
			}
		}
	}
	static WorkspacePool &get()
	{
		static WorkspacePool p;
		return p;
	}
	// finder workspace for blocks of up to max_n bytes with at least per_pos pool entries per byte
	MfWorkspace *take_mf(size_t max_n, double per_pos, int device, double *got_per_pos)
	{
		{
			std::lock_guard<std::mutex> lk(mu);
			size_t best = mf.size();
			for (size_t i = 0; i < mf.size(); i++)
				if (mf[i].device == device && mf[i].w->max_n >= max_n && mf[i].w->max_n <= max_n + max_n / 2 + (1u << 20) &&
				    mf[i].per_pos >= per_pos && (best == mf.size() || mf[i].w->max_n < mf[best].w->max_n))
					best = i;
			if (best != mf.size()) {
				MfWorkspace *w = mf[best].w;
				*got_per_pos = mf[best].per_pos;
				idle_bytes -= mf_bytes(w->max_n, mf[best].per_pos);
				mf.erase(mf.begin() + (long)best);
				return w;
			}
		}
		MfWorkspace *w = nullptr;
		std::lock_guard<std::mutex> al(DeviceBudget::alloc_mu());
		if (!DeviceBudget::make_room(mf_bytes(max_n, per_pos), device))
			return nullptr;
		if (mf_workspace_create(&w, max_n, per_pos) != 0) {
			mf_workspace_destroy(w);
			trim(device);
			DevicePool::get().trim(device);
			w = nullptr;
			if (mf_workspace_create(&w, max_n, per_pos) != 0) {
				mf_workspace_destroy(w);
				return nullptr;
			}
		}
		*got_per_pos = per_pos;
		return w;
	}
	void give_mf(MfWorkspace *w, double per_pos, int device)
	{
		if (!w)
			return;
		std::vector<MfEntry> m_out;
		std::vector<ScanEntry> s_out;
		{
			std::lock_guard<std::mutex> lk(mu);
			mf.push_back(MfEntry{w, device, per_pos});
			idle_bytes += mf_bytes(w->max_n, per_pos);
			evict_locked(m_out, s_out);
		}
		for (auto &e : m_out)
			mf_workspace_destroy(e.w);
		for (auto &e : s_out)
			scan_workspace_destroy(e.w);
	}
	ScanWorkspace *take_scan(int level, int64_t max_chunk, int device)
	{
		{
			std::lock_guard<std::mutex> lk(mu);
			for (size_t i = 0; i < scan.size(); i++)
				if (scan[i].device == device && scan[i].level == level && scan[i].max_chunk >= max_chunk &&
				    scan[i].max_chunk <= max_chunk + max_chunk / 2 + (1 << 20)) {
					ScanWorkspace *w = scan[i].w;
					idle_bytes -= scan_bytes(w);
					scan.erase(scan.begin() + (long)i);
					return w;
				}
		}
		ScanWorkspace *w = nullptr;
		std::lock_guard<std::mutex> al(DeviceBudget::alloc_mu());
		{
			// (what scan_workspace_create is going to ask for: candidate arrays for up to 2^28 positions, table, records)
			const size_t seg = (size_t)(max_chunk + 8192 < ((int64_t)1 << 28) ? max_chunk + 8192 : (int64_t)1 << 28);
			if (!DeviceBudget::make_room(seg * 24 + (size_t)(max_chunk / 31 + 16) * sizeof(MatchRec) + ((size_t)2 << 30), device))
				return nullptr;
		}
		if (scan_workspace_create(&w, level, max_chunk) != 0) {
			scan_workspace_destroy(w);
			trim(device);
			DevicePool::get().trim(device);
			w = nullptr;
			if (scan_workspace_create(&w, level, max_chunk) != 0) {
				scan_workspace_destroy(w);
				return nullptr;
			}
		}
		return w;
	}
	void give_scan(ScanWorkspace *w, int level, int64_t max_chunk, int device)
	{
		if (!w)
			return;
		std::vector<MfEntry> m_out;
		std::vector<ScanEntry> s_out;
		{
			std::lock_guard<std::mutex> lk(mu);
			scan.push_back(ScanEntry{w, device, level, max_chunk});
			idle_bytes += scan_bytes(w);
			evict_locked(m_out, s_out);
		}
		for (auto &e : m_out)
			mf_workspace_destroy(e.w);
		for (auto &e : s_out)
			scan_workspace_destroy(e.w);
	}
	void trim(int device = -1)
	{
		std::vector<MfEntry> m, mk;
		std::vector<ScanEntry> s, sk;
		{
			std::lock_guard<std::mutex> lk(mu);
			for (auto &e : mf)
				(device < 0 || e.device == device ? m : mk).push_back(e);
			for (auto &e : scan)
				(device < 0 || e.device == device ? s : sk).push_back(e);
			mf.swap(mk);
			scan.swap(sk);
			idle_bytes = 0;
			for (auto &e : mf)
				idle_bytes += mf_bytes(e.w->max_n, e.per_pos);
			for (auto &e : scan)
				idle_bytes += scan_bytes(e.w);
		}
		for (auto &e : m)
			mf_workspace_destroy(e.w);
		for (auto &e : s)
			scan_workspace_destroy(e.w);
	}
};

inline void pools_release_device(int device)
{
	WorkspacePool::get().trim(device);
	DevicePool::get().trim(device);
}


} // namespace lrzgpu
