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

// ---- device buffers ------------------------------------------------------------------------------
struct DevicePool {
	struct Entry {
		size_t cap;
		void *p;
		int device;
	};
	std::mutex mu;
	std::vector<Entry> idle;
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
				idle[best] = idle.back();
				idle.pop_back();
				return p;
			}
		}
		const size_t round = (size_t)2 << 20;
		*cap = (bytes + round - 1) / round * round;
		if (*cap == 0)
			*cap = round;
		void *p = nullptr;
		if (hipMalloc(&p, *cap) != hipSuccess) {
			(void)hipGetLastError();
			trim(device); // parked buffers of other shapes may be what is in the way
			if (hipMalloc(&p, *cap) != hipSuccess) {
				(void)hipGetLastError();
				return nullptr;
			}
		}
		return p;
	}
	void give(void *p, size_t cap, int device)
	{
		if (!p)
			return;
		std::lock_guard<std::mutex> lk(mu);
		idle.push_back(Entry{cap, p, device});
	}
	void trim(int device = -1)
	{
		std::vector<Entry> old, keep;
		{
			std::lock_guard<std::mutex> lk(mu);
			for (auto &e : idle)
				(device < 0 || e.device == device ? old : keep).push_back(e);
			idle.swap(keep);
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
	std::vector<MfEntry> mf;
	std::vector<ScanEntry> scan;
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
				mf[best] = mf.back();
				mf.pop_back();
				return w;
			}
		}
		MfWorkspace *w = nullptr;
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
		std::lock_guard<std::mutex> lk(mu);
		mf.push_back(MfEntry{w, device, per_pos});
	}
	ScanWorkspace *take_scan(int level, int64_t max_chunk, int device)
	{
		{
			std::lock_guard<std::mutex> lk(mu);
			for (size_t i = 0; i < scan.size(); i++)
				if (scan[i].device == device && scan[i].level == level && scan[i].max_chunk >= max_chunk &&
				    scan[i].max_chunk <= max_chunk + max_chunk / 2 + (1 << 20)) {
					ScanWorkspace *w = scan[i].w;
					scan[i] = scan.back();
					scan.pop_back();
					return w;
				}
		}
		ScanWorkspace *w = nullptr;
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
		std::lock_guard<std::mutex> lk(mu);
		scan.push_back(ScanEntry{w, device, level, max_chunk});
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
		}
		for (auto &e : m)
			mf_workspace_destroy(e.w);
		for (auto &e : s)
			scan_workspace_destroy(e.w);
	}
};

} // namespace lrzgpu
