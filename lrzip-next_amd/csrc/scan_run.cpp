// scan_run.cpp -- one whole-file compress run: readers bring the chunks into HBM, scanners run the rzip scan on them side
// by side and feed the literal blocks to the back end (pipeline.h) while they scan, the hash thread hashes the input, the
// calling thread commits the chunks in file order.  Reference: rzip_fd() / rzip_chunk(), src/rzip.c:586-1264.
#include "pipeline.h"

namespace lrzgpu {

// ---- one compress run ------------------------------------------------------------------------------
struct Run {
	lrzgpu_control *ctl;
	const CompressSource &in;
	CompressSink &out;
	const ChunkSelect *sel;
	Pipeline P;
	std::vector<std::unique_ptr<ChunkCtx>> chunks; // outlive every thread of the run
	std::vector<int> mine;                         // indices into `chunks` this run compresses, ascending
	std::mutex mu;
	std::condition_variable cv;
	size_t next_scan = 0;   // position in `mine` the next free scanner takes
	size_t committed = 0;   // chunks of `mine` already laid out: the reader stays a bounded distance ahead
	int scan_slots = 1;
	bool speculate = true;
	int64_t n_early = 0, n_violations = 0, n_rescans = 0;
	double t0 = 0;

	Run(lrzgpu_control *c, const CompressSource &i, CompressSink &o, const ChunkSelect *s) : ctl(c), in(i), out(o), sel(s) {}

	void fail(int e) { P.fail(e); } // (P.on_fail wakes this run's waiters)

	// ---- readers: chunk bytes into HBM ---------------------------------------------------------------------
	// With the input in HBM already one reader hands out views (or device-to-device copies).  A file or a host buffer
	// is read by as many readers as there are scanners, ALL of them on the same chunk: reader t of n takes the pieces
	// t, t + n, ... of every chunk, in file order.  One thread moves ~6 GB/s out of the page cache through its two
	// pinned pieces, and a scanner can only start on a chunk that is there completely (a match may run to the chunk's
	// end): read by one thread, the eighth chunk of the headline file was ready 2.7 s after the first -- and its scan
	// that much later; read by eight, chunk k is ready 45 ms after chunk k - 1.
	int n_readers = 1;
	struct ReadState { // per chunk of `mine`, guarded by mu
		int arrived = 0; // readers that have their pieces of the chunk in HBM
		int rc = 0;
		bool allocated = false;
	};
	std::vector<ReadState> read_state;
	// input copies the committer is through with and only the whole-input hash still reads (under mu)
	static constexpr int kHashHeldMax = 2;
	int hash_held_behind = 0;
	void reader_main(int t)
	{
		if (hipSetDevice(P.device) != hipSuccess) {
			fail(LRZGPU_E_HIP);
			return;
		}
		hipStream_t s = nullptr;
		RawBuf<uint8_t> stage_buf[2]; // pinned, from the pool (a run after the first finds them there)
		uint8_t *stage[2] = {nullptr, nullptr};
		hipEvent_t done[2] = {nullptr, nullptr};
		auto cleanup = [&] {
			// (a failing run arrives here with copies out of the pinned pieces still queued: the stream and the pieces go
			// back to their pools, where the next run -- or another run of this process -- takes them, only once it is idle)
			if (s)
				(void)stream_wait(s);
			for (int q = 0; q < 2; q++)
				if (done[q])
					(void)hipEventDestroy(done[q]);
			if (s)
				StreamPool::get().give(s);
		};
		if (make_stream(&s) != hipSuccess) {
			fail(LRZGPU_E_HIP);
			return;
		}
		const bool pieces = !in.dev && !in.dev_chunks; // host memory or a file: through pinned pieces
		if (pieces) {
			stage_buf[0].alloc(STAGE_BYTES, true);
			stage_buf[1].alloc(STAGE_BYTES, true);
			stage[0] = stage_buf[0].data();
			stage[1] = stage_buf[1].data();
			if (hipEventCreateWithFlags(&done[0], hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&done[1], hipEventDisableTiming) != hipSuccess) {
				fail(LRZGPU_E_HIP);
				cleanup();
				return;
			}
		}
		bool used[2] = {false, false};
		int k = 0;
		for (size_t m = 0; m < mine.size(); m++) {
			ChunkCtx *cc = chunks[(size_t)mine[m]].get();
			int rc = 0;
			{
				// at most scan_slots + 1 chunks ahead of the committer hold input copies -- plus, for a file, the copies
				// the committer has let go of and the whole-input hash has not passed yet (md5_main reads the chunks from
				// HBM): where the pipeline outruns the hash (stored blocks, the census, -n) those are bounded too, or a file
				// larger than the free HBM would end in LRZGPU_E_NOMEM instead of streaming through.  The
				// hash works in chunk order on chunks that are already in HBM, so it never waits for a reader held here.
				// The first reader to arrive sets the chunk's buffer up for all of them.
				std::unique_lock<std::mutex> lk(mu);
				cv.wait(lk, [&] { return P.err || (m < committed + (size_t)scan_slots + 1 && hash_held_behind <= kHashHeldMax); });
				if (P.err)
					break;
				ReadState &rs = read_state[m];
				if (!rs.allocated) {
					rs.allocated = true;
					const bool interior = in.dev && ((uintptr_t)(in.dev + cc->offset) & 15) == 0 && cc->offset + cc->size < in.n;
					if (interior)
						cc->d_in = in.dev + cc->offset; // interior chunk of a resident buffer: readable past its end
					else if (in.dev_chunks && !in.dev_chunks[cc->index] && cc->size)
						rs.rc = LRZGPU_E_PARAM; // a chunk this run was asked for but not given
					else if (!cc->in_buf.alloc((size_t)cc->size + 256, P.device))
						rs.rc = LRZGPU_E_NOMEM;
					else
						cc->d_in = cc->in_buf.p;
				}
				rc = rs.rc;
			}
			hipError_t e = hipSuccess;
			if (!rc && cc->in_buf.p) {
				if (!pieces) {
					// (a chunk handed over on its own has no readable bytes behind its end: it is copied next to padding)
					const uint8_t *from = in.dev ? in.dev + cc->offset : in.dev_chunks[cc->index];
					if (cc->size)
						e = hipMemcpyAsync(cc->in_buf.p, from, (size_t)cc->size, hipMemcpyDeviceToDevice, s);
				} else {
					// copy / pread of this reader's next piece under the DMA of its last one
					for (int64_t o = (int64_t)t * (int64_t)STAGE_BYTES; o < cc->size && !rc; o += (int64_t)n_readers * (int64_t)STAGE_BYTES, k ^= 1) {
						const size_t len = (size_t)(cc->size - o < (int64_t)STAGE_BYTES ? cc->size - o : (int64_t)STAGE_BYTES);
						if (used[k] && event_wait(done[k]) != hipSuccess) {
							rc = LRZGPU_E_HIP;
							break;
						}
						if (in.host)
							memcpy(stage[k], in.host + cc->offset + o, len);
						else if (pread_all(in.fd, stage[k], len, in.fd_base + cc->offset + o) != 0) {
							rc = LRZGPU_E_IO;
							break;
						}
						if (hipMemcpyAsync(cc->in_buf.p + o, stage[k], len, hipMemcpyHostToDevice, s) != hipSuccess ||
						    hipEventRecord(done[k], s) != hipSuccess)
							rc = LRZGPU_E_HIP;
						used[k] = true;
					}
				}
				if (!rc && e == hipSuccess && t == 0)
					e = hipMemsetAsync(cc->in_buf.p + cc->size, 0, 256, s);
				if (!rc && (e != hipSuccess || stream_wait(s) != hipSuccess))
					rc = LRZGPU_E_HIP;
			}
			if (rc) {
				fail(rc);
				break;
			}
			std::lock_guard<std::mutex> lk(mu);
			if (++read_state[m].arrived == n_readers) {
				cc->input_ready = true;
				cv.notify_all();
				if (tracing())
					fprintf(stderr, "lrzgpu reader: chunk %d (%lld bytes) in HBM at %.3f s\n", cc->index, (long long)cc->size, now_s() - t0);
			}
		}
		cleanup();
	}

	// ---- whole-input hash (the reference feeds it from cksumthread, src/rzip.c:564-584): MD5 unless
	// control->hash_code names another of hashes[] (src/main.c:64-79) ---------------------------------------
	uint8_t digest[64] = {0};
	const int hash_code = ctl->hash_code; // control->hash_code, src/rzip.c:943-950, 1195-1219
	// the hash of bytes that are in HBM: down in pinned pieces, piece k + 1 on its way while piece k is hashed (no host
	// CPU but the hashing itself: the DMA engine moves them)
	struct DeviceHashFeed {
		static constexpr size_t kPiece = (size_t)32 << 20;
		RawBuf<uint8_t> stage[2];
		hipStream_t s = nullptr;
		size_t pending = 0; // bytes of the piece on its way (in stage[k ^ 1] once waited for)
		int k = 0;
		double t_hash = 0, t_wait = 0;
		int open(int device)
		{
			if (hipSetDevice(device) != hipSuccess)
				return LRZGPU_E_HIP;
			stage[0].alloc(kPiece, true);
			stage[1].alloc(kPiece, true);
			return make_stream(&s) == hipSuccess ? 0 : LRZGPU_E_HIP;
		}
		// hashes what was on its way, after asking for the next piece (d == nullptr: nothing more to ask for)
		int step(Hasher &m, const uint8_t *d, size_t len)
		{
			if (pending && stream_wait_timed() != 0)
				return LRZGPU_E_HIP;
			const size_t have = pending;
			const int from = k;
			pending = 0;
			if (d && len) {
				k ^= 1;
				if (hipMemcpyAsync(stage[k].data(), d, len, hipMemcpyDeviceToHost, s) != hipSuccess)
					return LRZGPU_E_HIP;
				pending = len;
			}
			if (have) {
				const double ta = now_s();
				m.update(stage[from].data(), have);
				t_hash += now_s() - ta;
			}
			return 0;
		}
		int stream_wait_timed()
		{
			const double ta = now_s();
			const hipError_t e = stream_wait(s);
			t_wait += now_s() - ta;
			return e == hipSuccess ? 0 : -1;
		}
		int range(Hasher &m, const uint8_t *d, int64_t n, const std::atomic<int> &err)
		{
			for (int64_t o = 0; o < n && !err.load(); o += (int64_t)kPiece) {
				const int rc = step(m, d + o, (size_t)(n - o < (int64_t)kPiece ? n - o : (int64_t)kPiece));
				if (rc)
					return rc;
			}
			return 0;
		}
		int drain(Hasher &m) { return step(m, nullptr, 0); }
		void close()
		{
			if (s) {
				(void)stream_wait(s);
				StreamPool::get().give(s);
				s = nullptr;
			}
		}
	};
	// the hash of a range of the input file where the page cache holds it, through a mapping that moves along the file;
	// what cannot be mapped is read
	int hash_file_range(Hasher &m, int64_t from, int64_t n)
	{
		const size_t window = (size_t)256 << 20, piece = (size_t)32 << 20;
		const long pg = sysconf(_SC_PAGESIZE);
		std::vector<uint8_t> buf;
		for (int64_t o = 0; o < n && !P.error();) {
			const size_t len = (size_t)(n - o < (int64_t)window ? n - o : (int64_t)window);
			const int64_t file_off = in.fd_base + from + o, aligned = file_off / pg * pg;
			const size_t lead = (size_t)(file_off - aligned);
			void *mp = mmap(nullptr, len + lead, PROT_READ, MAP_SHARED, in.fd, (off_t)aligned);
			if (mp != MAP_FAILED) {
				(void)madvise(mp, len + lead, MADV_SEQUENTIAL);
				for (size_t q = 0; q < len && !P.error(); q += piece)
					m.update((const uint8_t *)mp + lead + q, len - q < piece ? len - q : piece);
				munmap(mp, len + lead);
			} else {
				buf.resize(piece);
				for (size_t q = 0; q < len && !P.error(); q += piece) {
					const size_t l2 = len - q < piece ? len - q : piece;
					if (pread_all(in.fd, buf.data(), l2, file_off + (int64_t)q) != 0)
						return LRZGPU_E_IO;
					m.update(buf.data(), l2);
				}
			}
			o += (int64_t)len;
		}
		return 0;
	}
	void md5_main()
	{
		std::unique_ptr<Hasher> hasher = make_hasher(hash_code);
		if (!hasher) {
			fail(LRZGPU_E_PARAM);
			return;
		}
		Hasher &m = *hasher;
		int rc = 0;
		if (in.host) {
			m.update(in.host, (size_t)in.n);
		} else if (in.n) {
			DeviceHashFeed feed;
			bool feed_open = false;
			if (in.dev) {
				rc = feed.open(P.device);
				feed_open = true;
				if (!rc)
					rc = feed.range(m, in.dev, in.n, P.err);
			} else {
				// a file.  The readers bring every chunk of this run into HBM for its scan: the hash takes it from there
				// (the chunk's copy stays until the hash has passed it), like an input that was in HBM from the start --
				// hashing out of the page cache, mapped or read, costs this thread the page faults or the memcpy of the
				// whole input, and this thread's speed is a floor of the run.  Chunks of the file that are not this
				// run's are hashed from the file.
				for (size_t c = 0; c < chunks.size() && !rc && !P.error(); c++) {
					ChunkCtx *cc = chunks[c].get();
					if (!cc->hash_holds) {
						if (feed_open)
							rc = feed.drain(m);
						if (!rc)
							rc = hash_file_range(m, cc->offset, cc->size);
						continue;
					}
					{
						std::unique_lock<std::mutex> lk(mu);
						cv.wait(lk, [&] { return P.err || cc->input_ready; });
						if (P.err)
							break;
					}
					if (!feed_open) {
						rc = feed.open(P.device);
						feed_open = true;
					}
					if (!rc)
						rc = feed.range(m, cc->d_in, cc->size, P.err);
					if (!rc)
						rc = feed.drain(m); // (the chunk's last piece is on the host before the copy may go)
					std::lock_guard<std::mutex> lk(mu);
					cc->hash_holds = false;
					if (cc->release_wanted) {
						cc->in_buf.release();
						cc->d_in = nullptr;
						hash_held_behind--;
						cv.notify_all();
					}
				}
			}
			if (feed_open) {
				if (!rc)
					rc = feed.drain(m);
				if (tracing())
					fprintf(stderr, "lrzgpu hash thread: %.2f s hashing, %.2f s waiting for the next piece from the device; done at %.2f s\n", feed.t_hash,
						feed.t_wait, now_s() - t0);
				feed.close();
			}
			if (rc) {
				fail(rc);
				return;
			}
		}
		m.finish(digest);
		t_hash_done = now_s();
	}
	double t_hash_done = 0; // when the whole-input hash was through (md5_main)

	// ---- one chunk through K1..K5 with early block release ---------------------------------------
	struct Scanner {
		Feeder F;
		ScanWorkspace *sw = nullptr;
		DevBuf runs;
		explicit Scanner(Pipeline &p) : F(p) {}
	};

	// Resolvers of different chunks must not share a CU: each is ONE latency-bound wavefront, and the dispatcher
	// happily packs eight single-wave workgroups onto the same SIMDs (measured: the first scan segment takes
	// 653 ms with eight resolvers side by side against 413 ms alone, with nothing else on the GPU).  Scanner k gets
	// the CUs k, k + 8, k + 16, ... for its scan stream: disjoint sets whatever the mask-bit-to-XCD mapping is (one
	// XCD each if the bits go round the XCDs), 32 CUs wide so that the K1 kernels on the same stream keep their
	// bandwidth.  Such streams are blocking streams: nothing in the pipeline uses the null stream.
	std::atomic<int> scanner_ids{0};
	hipError_t make_scan_stream(hipStream_t *s)
	{
		hipDeviceProp_t prop;
		if (scan_slots > 1 && hipGetDeviceProperties(&prop, P.device) == hipSuccess && prop.multiProcessorCount >= 64) {
			const int ncu = prop.multiProcessorCount > 256 ? 256 : prop.multiProcessorCount;
			const int k = scanner_ids.fetch_add(1) % 8;
			const int kind = 16 + k;
			if ((*s = StreamPool::get().take(P.device, kind)) != nullptr)
				return hipSuccess;
			uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
			for (int c = k; c < ncu; c += 8)
				mask[c >> 5] |= 1u << (c & 31);
			if (hipExtStreamCreateWithCUMask(s, (uint32_t)((ncu + 31) / 32), mask) == hipSuccess) {
				StreamPool::get().created(*s, P.device, kind);
				return hipSuccess;
			}
			(void)hipGetLastError();
		}
		return make_stream(s, true);
	}

	int scanner_open(Scanner &S)
	{
		if (hipSetDevice(P.device) != hipSuccess || make_scan_stream(&S.F.ms) != hipSuccess)
			return LRZGPU_E_HIP;
		const int ngate = scan_slots > 1 ? 2 : 6;
		for (int k = 0; k < ngate; k++) {
			hipStream_t gs;
			if (make_stream(&gs) != hipSuccess)
				return LRZGPU_E_HIP;
			S.F.gate_streams.push_back(gs);
		}
		const int64_t cap_chunk = P.sz.max_chunk < in.n ? P.sz.max_chunk : in.n;
		S.sw = WorkspacePool::get().take_scan(P.sz.rzip_level, cap_chunk, P.device);
		return S.sw ? 0 : LRZGPU_E_NOMEM;
	}
	void scanner_close(Scanner &S)
	{
		S.F.destroy();
		const int64_t cap_chunk = P.sz.max_chunk < in.n ? P.sz.max_chunk : in.n;
		WorkspacePool::get().give_scan(S.sw, P.sz.rzip_level, cap_chunk, P.device);
		S.sw = nullptr;
		S.runs.release();
	}

	// gathers stream-1 bytes [S0, S1) from `runs` (absolute dst offsets)
	int gather(Scanner &S, ChunkCtx *cc, const std::vector<CopyRun> &runs, int64_t S0, int64_t S1)
	{
		if (runs.empty() || S1 <= S0)
			return 0;
		hipStream_t ms = S.F.ms;
		if (runs.size() * sizeof(CopyRun) > S.runs.cap && !S.runs.alloc((runs.size() * 2 + 64) * sizeof(CopyRun), P.device))
			return LRZGPU_E_NOMEM;
		if (hipMemcpyAsync(S.runs.p, runs.data(), runs.size() * sizeof(CopyRun), hipMemcpyHostToDevice, ms) != hipSuccess)
			return LRZGPU_E_HIP;
		EventTimer tg(ms);
		int gr = gather_runs_device(cc->d_in, cc->stream1.p, (const CopyRun *)S.runs.p, (int)runs.size(), S0, S1, ms);
		tg.stop();
		if (gr != 0 || stream_wait(ms) != hipSuccess)
			return LRZGPU_E_HIP;
		ProfileStore &ps = ProfileStore::get();
		std::lock_guard<std::mutex> lk(ps.mu);
		ps.p.gather_ms += tg.ms_noted(ps, PK_GATHER);
		ps.p.gather_launches++;
		ps.p.gather_bytes += S1 - S0;
		return 0;
	}

	static bool census_allowed()
	{
		const char *e = getenv("LRZGPU_CENSUS"); // 0: every chunk through the resolver (tests compare both ways)
		return !(e && *e == '0');
	}
	// ---- what a chunk's scan has handed to the back end so far (speculative early emission while the scan runs) ----
	struct ScanEmit {
		int64_t E = 0;          // chunk position up to which stream-1 bytes have been gathered
		int64_t Sg = 0;         // stream-1 bytes gathered so far
		int64_t seen = 0;       // match records consumed
		int64_t blocks_out = 0; // full stream-1 blocks already submitted
		bool violated = false;
		std::map<int64_t, Job *> early; // stream-1 offset -> job
		std::set<Job *> starting;       // early jobs whose block is not complete yet
		std::vector<MatchRec> rec_buf;
		bool can_early = false; // early start of blocks: LZMA blocks only, and not under a filter (a block is filtered once, in place, whole)
	};
#define SCAN_EMIT_LOCALS(em)                                                                                                   \
	int64_t &E = em.E, &Sg = em.Sg, &seen = em.seen, &blocks_out = em.blocks_out;                                          \
	bool &violated = em.violated;                                                                                          \
	std::map<int64_t, Job *> &early = em.early;                                                                            \
	std::set<Job *> &starting = em.starting;                                                                               \
	std::vector<MatchRec> &rec_buf = em.rec_buf;                                                                           \
	const bool can_early = em.can_early;                                                                                   \
	const int64_t chunk_size = cc->size, bufsize = P.sz.stream_bufsize;                                                    \
	Feeder &F = S.F;                                                                                                       \
	(void)E, (void)Sg, (void)seen, (void)blocks_out, (void)violated, (void)early, (void)starting, (void)rec_buf, (void)can_early, (void)chunk_size, \
		(void)bufsize, (void)F
	void make_early(Job *j)
	{
		std::lock_guard<std::mutex> lk(P.mu);
		j->early = true;
		P.early_unclaimed++;
		P.n_early_jobs++;
	}
	// the scan has got as far as `upto` (or is through: final_call): the records it made since the last call become
	// copy runs, the literal bytes that are decided are gathered, every stream-1 block they complete goes to the back
	// end, the block under construction is staged for an early start
	int advance(Scanner &S, ChunkCtx *cc, ScanEmit &em, const ScanState &h, int64_t upto, bool final_call, const std::vector<MatchRec> *final_recs)
	{
		SCAN_EMIT_LOCALS(em);
		const int64_t nrec = final_recs ? (int64_t)final_recs->size() : h.n_records;
		std::vector<CopyRun> runs;
		const int64_t S_before = Sg;
		if (nrec > seen) {
			const MatchRec *rp;
			if (final_recs)
				rp = final_recs->data() + seen;
			else {
				rec_buf.resize((size_t)(nrec - seen));
				if (d2h_pageable(rec_buf.data(), S.sw->records + seen, (size_t)(nrec - seen) * sizeof(MatchRec), F.ms) != hipSuccess)
					return LRZGPU_E_HIP;
				rp = rec_buf.data();
			}
			for (int64_t k = 0; k < nrec - seen && !violated; k++) {
				const MatchRec &r = rp[k];
				if (r.p < E) {
					violated = true; // a match reaches back over bytes already emitted as literals
					break;
				}
				if (E < r.p) {
					runs.push_back(CopyRun{E, Sg, r.p - E});
					Sg += r.p - E;
				}
				E = r.p + r.len;
			}
			seen = nrec;
		}
		if (violated)
			return 0;
		int64_t Fp = final_call ? chunk_size : upto - spec_margin();
		if (!final_call && h.cur_len > 0 && h.cur_p < Fp)
			Fp = h.cur_p;
		if (Fp > chunk_size)
			Fp = chunk_size;
		if (Fp > E) {
			if (!runs.empty() && runs.back().src_off + runs.back().len == E)
				runs.back().len += Fp - E;
			else
				runs.push_back(CopyRun{E, Sg, Fp - E});
			Sg += Fp - E;
			E = Fp;
		}
		if (Sg > S_before) {
			int g = gather(S, cc, runs, S_before, Sg);
			if (g)
				return g;
		}
		if (final_call)
			return 0;
		std::vector<Job *> fresh;
		while ((blocks_out + 1) * bufsize <= Sg) {
			const int64_t off = blocks_out * bufsize;
			auto it = early.find(off);
			Job *j = it != early.end() ? it->second : nullptr; // started while it was being filled: now whole
			if (j)
				starting.erase(j);
			else {
				j = F.new_job(cc, BlockRef{1, off, bufsize});
				// encoders with nothing to do: the finder hands them a first part of the block at once
				if (can_early && P.early_split && P.want_early())
					make_early(j);
				early[off] = j;
			}
			fresh.push_back(j);
			blocks_out++;
		}
		{
			std::lock_guard<std::mutex> lk(mu);
			n_early += (int64_t)fresh.size();
		}
		int sr2 = F.submit(fresh);
		if (sr2)
			return sr2;
		// the block under construction: once a first part of it is there and encoders have
		// nothing to do, the finder runs on what is there and an encoder starts on those lists; every further
		// piece is another run on the longer prefix
		if (can_early) {
			const int64_t off = blocks_out * bufsize, have = Sg - off;
			auto it = early.find(off);
			Job *ej = it != early.end() ? it->second : nullptr;
			if (!ej && have >= P.early_first && P.want_early()) {
				ej = F.new_job(cc, BlockRef{1, off, bufsize});
				make_early(ej);
				early[off] = ej;
				starting.insert(ej);
			}
			if (ej)
				P.stage(ej, have);
		}
		if (P.error())
			return P.error();
		return F.poll(false);
	}

	int scan_chunk(Scanner &S, ChunkCtx *cc, int64_t vr_in)
	{
		ScanEmit em;
		em.can_early = speculate && P.early_mode != 0 && !P.sz.zstd && !P.sz.no_compress && !P.filter_flag && P.sz.stream_bufsize >= 4096;
		SCAN_EMIT_LOCALS(em);
		cc->vr_in = vr_in;
		cc->file_order.clear();
		cc->stream0.clear();
		if (!cc->stream1.p && !cc->stream1.alloc((size_t)chunk_size + 256, P.device))
			return LRZGPU_E_NOMEM;
		{
			int pr = F.poll(true); // nothing of an earlier chunk may still sit in the descriptor arena
			if (pr)
				return pr;
			int rr = F.reserve((size_t)(chunk_size / bufsize + 8) * 2);
			if (rr)
				return rr;
		}
		ScanProgressFn progress = nullptr;
		if (speculate)
			progress = [&](const ScanState &h, int64_t upto) -> int { return advance(S, cc, em, h, upto, false, nullptr); };

		ScanResult sr;
		int64_t vr = vr_in;
		// (the file's last chunk: nobody needs the victim_round it ends with, so it may skip the resolver when it holds
		// no repeat at all -- rzip_census.hip)
		int r = scan_chunk_device(S.sw, cc->d_in, chunk_size, P.sz.rzip_level, &vr, &sr, F.ms, progress, cc->last && census_allowed());
		if (r)
			return r < -50 ? r : (r == -4 ? LRZGPU_E_NOMEM : LRZGPU_E_INTERNAL);
		{
			std::lock_guard<std::mutex> lk(mu); // scanners read a predecessor's vr_out under mu (predicted_vr)
			cc->vr_out = vr;
		}
		EmitResult er;
		emit_streams(sr.records, chunk_size, cc->chunk_bytes, sr.crc, &er);
		cc->stream0.swap(er.stream0);
		cc->stream1_len = er.stream1_len;
		if (speculate && !violated) {
			int a = advance(S, cc, em, sr.final_state, chunk_size, true, &sr.records);
			if (a)
				return a;
		}
		if (!speculate || violated || Sg != er.stream1_len) {
			// (re)build stream 1 from the final run table; early blocks, if any, are void
			if (speculate && violated) {
				ProfileStore &ps = ProfileStore::get();
				std::lock_guard<std::mutex> lk(ps.mu);
				ps.p.spec_rollbacks++;
			}
			if (!early.empty()) {
				{
					std::lock_guard<std::mutex> lk(mu);
					n_violations++;
				}
				{
					ProfileStore &ps = ProfileStore::get();
					std::lock_guard<std::mutex> lk(ps.mu);
					ps.p.spec_cancelled_blocks += (int64_t)early.size();
				}
				std::vector<Job *> dead;
				for (auto &kv : early)
					dead.push_back(kv.second);
				for (Job *j : dead)
					j->cancelled = true;
				int pr = F.poll(true);
				if (pr)
					return pr;
				P.cancel_and_wait(dead);
				if (P.error())
					return P.error();
				early.clear();
			}
			int g = gather(S, cc, er.runs, 0, er.stream1_len);
			if (g)
				return g;
		}
		if (hipMemsetAsync(cc->stream1.p + er.stream1_len, 0, 256, F.ms) != hipSuccess || stream_wait(F.ms) != hipSuccess)
			return LRZGPU_E_HIP;
		return finish_chunk_blocks(S, cc, em);
	}

	// the chunk's blocks in the order the reference flushes them; blocks started early are reused
	int finish_chunk_blocks(Scanner &S, ChunkCtx *cc, ScanEmit &em)
	{
		SCAN_EMIT_LOCALS(em);
		std::vector<BlockRef> refs;
		block_order(cc->stream0, cc->chunk_bytes, cc->stream1_len, bufsize, &refs);
		std::vector<Job *> fresh;
		for (const BlockRef &br : refs) {
			Job *j = nullptr;
			if (br.streamno == 1 && br.len == bufsize) {
				auto it = early.find(br.off);
				if (it != early.end()) {
					j = it->second;
					early.erase(it);
					if (starting.erase(j))
						fresh.push_back(j); // started early, completed by the last piece of the scan: the whole block now
				}
			}
			if (!j) {
				j = F.new_job(cc, br);
				fresh.push_back(j);
			}
			cc->file_order.push_back(j);
		}
		if (!early.empty()) {
			// a block started early that the chunk's last, shorter block took the place of: withdrawn (every other
			// early block is a full stream-1 block of the final layout)
			std::vector<Job *> dead;
			for (auto &kv : early) {
				if (!starting.count(kv.second))
					return LRZGPU_E_INTERNAL;
				dead.push_back(kv.second);
			}
			P.cancel_and_wait(dead);
			if (P.error())
				return P.error();
			early.clear();
		}
		return F.submit(fresh);
	}

	void scanner_main()
	{
		Scanner S(P);
		int rc = scanner_open(S);
		while (!rc) {
			ChunkCtx *cc = nullptr;
			int64_t vr_in = 0;
			{
				std::unique_lock<std::mutex> lk(mu);
				if (P.err || next_scan >= mine.size())
					break;
				const size_t m = next_scan++;
				cc = chunks[(size_t)mine[m]].get();
				cv.wait(lk, [&] { return P.err || cc->input_ready; });
				if (P.err)
					break;
				vr_in = predicted_vr(cc->index);
			}
			rc = scan_chunk(S, cc, vr_in);
			if (!rc)
				rc = S.F.poll(true);
			if (rc)
				break;
			std::lock_guard<std::mutex> lk(mu);
			cc->scanned = true;
			cc->t_scanned = now_s();
			cv.notify_all();
		}
		if (rc)
			fail(rc);
		scanner_close(S);
	}

	// victim_round a chunk should start from (mu held): what its predecessor left if that is known, the
	// caller's hint for the first chunk of a partial run, else 0 -- the value only moves when a tag value
	// collects max_chain_len table entries, which ordinary data does rarely
	int64_t predicted_vr(int index)
	{
		if (index == 0)
			return 0;
		if (sel && sel->victim_in && sel->victim_in[index] >= 0)
			return sel->victim_in[index]; // the caller's word comes first (lrzgpu.h: "gives the value chunk k starts from")
		const ChunkCtx *prev = chunks[(size_t)index - 1].get();
		if (prev->scanned)
			return prev->vr_out;
		return 0;
	}

	struct Commit;
	int setup();
	int rescan_if_the_guess_was_wrong(ChunkCtx *cc, Commit &c);
	int commit_chunk(size_t m, Commit &c);
	int run();
};

// sizing, the chunks of the file, how many workers of each kind (everything before the first thread)
int Run::setup()
{
	int rc = select_device(ctl->device);
	if (rc)
		return rc;
	P.ctl = ctl;
	P.device = ctl->device;
	rc = sizing_for_input(ctl, in.n, &P.sz);
	if (rc)
		return rc;
	if (control_filter(ctl, &P.filter_flag, &P.filter_delta) || hash_length(hash_code) < 0)
		return LRZGPU_E_PARAM;
	// host encoders: as asked, else the -p threads capped by the CPUs this process can really use
	// (more runnable threads than the cgroup quota only buys throttling)
	P.n_encoders = ctl->host_threads > 0 ? ctl->host_threads : (ctl->threads > 0 ? ctl->threads : 1);
	if (ctl->host_threads <= 0) {
		const int usable = usable_cpus();
		if (P.n_encoders > usable)
			P.n_encoders = usable;
	}
	if (P.sz.zstd && !ZstdLib::get().ok)
		return LRZGPU_E_PARAM; // --zstd asked for and no libzstd.so.1 on this host
	P.n_gpu_workers = ctl->gpu_slots > 0 ? ctl->gpu_slots : 3;
	P.held_limit = (size_t)P.n_encoders + (size_t)P.n_gpu_workers + 2;
	ctl->stream_bufsize = P.sz.stream_bufsize;
	ctl->dictSize_used = P.sz.dict_size;
	ctl->threads_used = P.sz.threads;
	ctl->st_size = in.n;
	speculate = true; // (blocks are released to the back end while their chunk is still being scanned)
	// early start of blocks.  LRZGPU_EARLY_START: 0 off, 1 (default) while encoders have nothing to
	// do, 2 every block (tests); LRZGPU_EARLY_STEP: bytes of a block between two finder runs (default 1/16 of a block).
	// None of it changes the output.
	{
		const char *e = getenv("LRZGPU_EARLY_START"); // read per run: tests flip it inside one process
		P.early_mode = e ? atoi(e) : 1;
		if (P.early_mode < 0 || P.early_mode > 2)
			P.early_mode = 1;
		int64_t step = P.sz.stream_bufsize / 16;
		if (const char *t = getenv("LRZGPU_EARLY_STEP"))
			if (atoll(t) > 0)
				step = atoll(t);
		if (step < 4096)
			step = 4096;
		P.early_step = step;
		P.early_first = step;
		P.early_split = true;
	}

	// the chunks of the file (src/rzip.c:1041: at least one pass, even for an empty input; STDIN mode: one more,
	// empty, when the input ends exactly where a chunk does)
	{
		std::vector<int64_t> sizes;
		chunk_sizes_for(ctl, P.sz, in.n, &sizes);
		int64_t offset = 0;
		for (size_t k = 0; k < sizes.size(); k++) {
			std::unique_ptr<ChunkCtx> cc(new ChunkCtx());
			cc->index = (int)k;
			cc->offset = offset;
			cc->size = sizes[k];
			cc->chunk_bytes = chunk_bytes_for(cc->size);
			offset += cc->size;
			cc->last = k + 1 == sizes.size();
			chunks.push_back(std::move(cc));
		}
	}
	for (size_t k = 0; k < chunks.size(); k++)
		if (!sel || (sel->stride > 0 && (int)k % sel->stride == sel->first))
			mine.push_back((int)k);
	scan_slots = ctl->scan_slots > 0 ? ctl->scan_slots : 8;
	if ((size_t)scan_slots > mine.size())
		scan_slots = mine.empty() ? 1 : (int)mine.size();
	// Every GPU worker owns a finder workspace of ~240 bytes per byte of the largest block, for the whole run: with the
	// 134 MB blocks of a 32 GiB chunk that is 32 GB each, and eight of them beside the chunk (input copy + stream 1) do
	// not fit 288 GB.  So: as many workers as fit what the device has left beside the chunks in flight (at least one; the
	// output does not depend on the number).  Parked pool memory counts as free (it is given back on demand).
	if (!P.sz.zstd && !P.sz.no_compress && in.n > 0) {
		size_t in_flight = 0;
		for (int k : mine) {
			const size_t c = (size_t)chunks[(size_t)k]->size;
			if (c > in_flight)
				in_flight = c;
		}
		// per scanner: stream 1 of the chunk, a copy of it unless the caller's device buffer can be used in place, ~4 GB
		// of scan workspace
		in_flight = (size_t)(scan_slots + (in.dev ? 0 : 1)) * (2 * in_flight + ((size_t)4 << 30));
		const size_t avail = DeviceBudget::free_now() + WorkspacePool::get().idle_bytes + DevicePool::get().idle_bytes;
		size_t per_ws = WorkspacePool::mf_bytes((size_t)P.sz.stream_bufsize, P.mf_per_pos);
		if (avail != ~(size_t)0 && per_ws > ((size_t)1 << 30)) { // (small blocks: nothing to bound)
			const size_t room = avail > in_flight + DeviceBudget::margin() ? avail - in_flight - DeviceBudget::margin() : 0;
			// One workspace must fit.  Its list pools are sized for 16 entries per block byte (text needs ~5, and a
			// finder run that outgrows its pool is repeated with a larger one): a block too large for that gets what
			// fits, down to 4 entries per byte; below that the block is beyond this device -- said now, not as an
			// out-of-memory error minutes into the run (the ceiling: lrzgpu_max_block_bytes()).
			while (per_ws > room && P.mf_per_pos > kMinPoolPerPos) {
				P.mf_per_pos = P.mf_per_pos > 8 ? P.mf_per_pos - 4 : P.mf_per_pos - 2;
				per_ws = WorkspacePool::mf_bytes((size_t)P.sz.stream_bufsize, P.mf_per_pos);
			}
			if (per_ws > room) {
				if (ctl->verbose || getenv("LRZGPU_TRACE"))
					fprintf(stderr, "lrzgpu: blocks of %lld bytes need a match-finder workspace of %zu MiB; %zu MiB are free beside the chunks in flight\n",
						(long long)P.sz.stream_bufsize, per_ws >> 20, room >> 20);
				return LRZGPU_E_BLOCK_TOO_LARGE;
			}
			size_t fit = room / per_ws;
			if (fit < 1)
				fit = 1;
			if ((size_t)P.n_gpu_workers > fit) {
				if (getenv("LRZGPU_TRACE"))
					fprintf(stderr, "lrzgpu driver: %d GPU workers asked for, %zu finder workspaces of %zu MiB fit beside the chunks in flight\n",
						P.n_gpu_workers, fit, per_ws >> 20);
				P.n_gpu_workers = (int)fit;
				P.held_limit = (size_t)P.n_encoders + (size_t)P.n_gpu_workers + 2;
			}
		}
	}
	if (ctl->verbose)
		fprintf(stderr, "lrzgpu: threads %d bufsize %lld dict %u chunk %lld chunks %zu (%zu here) scanners %d encoders %d gpu workers %d\n",
			P.sz.threads, (long long)P.sz.stream_bufsize, P.sz.dict_size, (long long)P.sz.max_chunk, chunks.size(), mine.size(),
			scan_slots, P.n_encoders, P.n_gpu_workers);

	return 0;
}

// ---- the committer's side (the calling thread): chunks in file order --------------------------------------------
struct Run::Commit {
	std::unique_ptr<Scanner> rescanner;
	double t_scan_last = 0, t_blocks = 0;
};

// the chunk was scanned from a guessed victim_round: if its predecessor left another value, its blocks are void and it is
// scanned again, here, from the right one
int Run::rescan_if_the_guess_was_wrong(ChunkCtx *cc, Commit &c)
{
	const ChunkCtx *prev = chunks[(size_t)cc->index - 1].get();
	if (prev->vr_out != cc->vr_in) {
		// the chunk was scanned from the wrong victim_round: void its blocks and scan it again
		n_rescans++;
		std::vector<Job *> dead;
		for (auto &j : cc->jobs)
			dead.push_back(j.get());
		P.cancel_and_wait(dead);
		if (P.error()) 
			return P.error();
		if (!c.rescanner) {
			c.rescanner.reset(new Scanner(P));
			int orc = scanner_open(*c.rescanner);
			if (orc) 
				return orc;
		}
		{
			std::lock_guard<std::mutex> lk(mu);
			cc->scanned = false; // its vr_out is not to be trusted while it is scanned again
		}
		int src = scan_chunk(*c.rescanner, cc, prev->vr_out);
		if (!src)
			src = c.rescanner->F.poll(true);
		if (src) 
			return src;
		{
			std::lock_guard<std::mutex> lk(mu);
			cc->scanned = true;
			cv.notify_all();
		}
	}
	return 0;
}

// chunk m of this run: wait for its scan (and the check of its victim_round guess), for its blocks, lay it out, hand it on
int Run::commit_chunk(size_t m, Commit &c)
{
	int ret = 0;
	ChunkCtx *cc = chunks[(size_t)mine[m]].get();
	{
		std::unique_lock<std::mutex> lk(mu);
		cv.wait(lk, [&] { return P.err || cc->scanned; });
		if (P.err) 
			return P.err;
	}
	// victim_round chain: only checkable when this run also scanned the predecessor
	if (cc->index > 0 && (!sel || sel->stride == 1)) {
		const int rr = rescan_if_the_guess_was_wrong(cc, c);
		if (rr)
			return rr;
	}
	c.t_scan_last = cc->t_scanned;
	{
		// no rescan can be asked for any more: the input copy goes (now, or when the hash has passed it)
		std::lock_guard<std::mutex> lk(mu);
		if (cc->hash_holds) {
			cc->release_wanted = true;
			hash_held_behind++; // (readers wait while more than kHashHeldMax copies are held for the hash alone)
		} else {
			cc->in_buf.release();
			cc->d_in = nullptr;
		}
	}
	// wait for every block of the chunk (discarded early ones included: they reference its buffers)
	{
		std::unique_lock<std::mutex> lk(P.mu);
		P.cv_done.wait(lk, [&] {
			if (P.err)
				return true;
			for (auto &j : cc->jobs)
				if (!j->finished)
					return false;
			return true;
		});
		if (P.err) 
			return P.err;
	}
	c.t_blocks = now_s();
	cc->stream1.release();
	// ordered container assembly of this chunk, straight into the sink's memory where it offers some
	std::vector<DoneBlock> blocks;
	for (Job *j : cc->file_order)
		blocks.push_back(std::move(j->done));
	const size_t total = chunk_image_size(cc->chunk_bytes, blocks);
	uint8_t *space = (sel && sel->on_chunk) ? nullptr : out.append_space(total);
	if (space) {
		write_chunk_raw(space, cc->chunk_bytes, cc->last, cc->size, blocks);
	} else {
		std::unique_ptr<uint8_t, void (*)(void *)> img((uint8_t *)malloc(total ? total : 1), free);
		if (!img)
			ret = LRZGPU_E_NOMEM;
		else {
			write_chunk_raw(img.get(), cc->chunk_bytes, cc->last, cc->size, blocks);
			if (sel && sel->on_chunk) {
				if (sel->on_chunk(sel->ctx, cc->index, cc->vr_in, cc->vr_out, img.get(), (int64_t)total) != 0)
					ret = LRZGPU_E_IO;
			} else if (out.put(img.get(), total) != 0)
				ret = LRZGPU_E_IO;
		}
	}
	cc->jobs.clear();
	cc->file_order.clear();
	std::vector<uint8_t>().swap(cc->stream0);
	std::lock_guard<std::mutex> lk(mu);
	committed = m + 1;
	cv.notify_all();
	return ret;
}

int Run::run()
{
	int ret = setup();
	if (ret)
		return ret;
	t0 = now_s();
	g_trace_t0 = t0;
	g_trace_events.store((getenv("LRZGPU_TRACE") && atoi(getenv("LRZGPU_TRACE")) >= 2) ? 1 : 0, std::memory_order_relaxed);
	P.on_fail = [this] {
		std::lock_guard<std::mutex> lk(mu);
		cv.notify_all();
	};
	const bool want_md5 = !sel || sel->with_md5;
	if (in.dev_chunks && (want_md5 || !sel))
		return LRZGPU_E_PARAM; // the whole-input hash needs the whole input (checked before any thread exists)
	if (want_md5 && !in.host && !in.dev && !in.dev_chunks)
		for (int k : mine)
			chunks[(size_t)k]->hash_holds = true; // a file: the hash reads this run's chunks from their copies in HBM (md5_main)
	P.start();
	std::vector<std::thread> side;
	if (want_md5)
		side.emplace_back([this] { P.guarded([this] { md5_main(); }, 3); });
	n_readers = (in.dev || in.dev_chunks) ? 1 : std::max(1, std::min(scan_slots, 8));
	read_state.assign(mine.size(), ReadState());
	for (int t = 0; t < n_readers; t++)
		side.emplace_back([this, t] { P.guarded([this, t] { reader_main(t); }, 4); });
	for (int k = 0; k < scan_slots; k++)
		side.emplace_back([this] { P.guarded([this] { scanner_main(); }, 2); });

	// ---- committer: chunks in file order ------------------------------------------------------------
	const bool whole_file = !sel;
	if (whole_file && out.begin(21) != 0) // magic placeholder (compress_file, src/lrzip.c:1487-1547)
		ret = LRZGPU_E_IO;
	Commit c;
	for (size_t m = 0; m < mine.size() && !ret; m++)
		ret = commit_chunk(m, c);
	if (ret)
		fail(ret);
	if (c.rescanner)
		scanner_close(*c.rescanner);
	// every thread ends on its own (work done) or on the error flag; chunks and jobs outlive them all
	for (auto &t : side)
		t.join();
	{
		// blocks still in flight after a failure reference chunk buffers: wait them out before those go
		std::unique_lock<std::mutex> lk(P.mu);
		if (!P.err)
			P.cv_done.wait(lk, [&] {
				for (auto &c : chunks)
					for (auto &j : c->jobs)
						if (!j->finished)
							return false;
				return true;
			});
	}
	P.stop();
	const double t_md5 = now_s();
	if (!ret && P.err)
		ret = P.err;
	if (ret)
		return ret;

	if (whole_file) {
		// the hash after the last chunk (none for code 0, "CRC": the chunk CRCs are all there is) and its code in magic[14]
		const int hash_len = hash_code == 0 ? 0 : hash_length(hash_code);
		if (hash_len > 0 && out.put(digest, (size_t)hash_len) != 0)
			return LRZGPU_E_IO;
		uint8_t magic[21];
		write_magic_for(magic, ctl, P.sz, in.n, chunks.size());
		if (out.finish(magic, 21) != 0)
			return LRZGPU_E_IO;
	}
	if (want_md5) {
		memcpy(ctl->hash_resblock, digest, 16);
		memcpy(ctl->hash_full, digest, sizeof(ctl->hash_full));
	}
	if (tracing())
		fprintf(stderr, "lrzgpu driver: %zu chunks, %d scanners: last scan done %.2f  last finder %.2f  last encode %.2f  all blocks %.2f  md5 joined %.2f  assembled %.2f s (since start); early blocks %lld, redone chunks %lld, rescans (victim_round) %lld; worker sums: block copy+gate %.2f finder %.2f lists D2H %.2f, encoders busy %.2f idle %.2f s\n",
			chunks.size(), scan_slots, c.t_scan_last - t0, P.t_last_mf - t0, P.t_last_enc - t0, c.t_blocks - t0, t_md5 - t0, now_s() - t0,
			(long long)n_early, (long long)n_violations, (long long)n_rescans, P.blk_busy, P.mf_busy, P.d2h_busy, P.enc_busy, P.enc_wait);
	{
		ProfileStore &ps = ProfileStore::get();
		std::lock_guard<std::mutex> lk(ps.mu);
		ps.p.victim_rescans += n_rescans;
		const double vals[8] = {P.enc_busy, P.enc_wait, P.mf_busy, P.d2h_busy, c.t_scan_last - t0, P.t_last_mf - t0, P.t_last_enc - t0, now_s() - t0};
		for (int k = 0; k < 8; k++)
			ps.p.pipeline_s[k] += vals[k] > 0 ? vals[k] : 0;
		ps.p.early_s[0] += P.t_first_enc > t0 ? P.t_first_enc - t0 : 0;
		ps.p.early_s[1] += P.rest_wait;
		ps.p.early_s[2] += (double)P.n_early_jobs;
		ps.p.early_s[3] += (double)P.n_early_stages;
		ps.p.shard_s[4] += t_hash_done > t0 ? t_hash_done - t0 : 0;
	}
	{
		LzmaParams p;
		if (!P.sz.no_compress && lzma_normalize(p, P.sz.level, P.sz.dict_size, 3, 0, 2, P.sz.level < 7 ? 32 : 64) == LZ_OK)
			lzma_write_props(p, ctl->lzma_properties);
	}
	return 0;
}

int run_compress(lrzgpu_control *ctl, const CompressSource &in, CompressSink &out, const ChunkSelect *sel)
{
	try {
		Run r(ctl, in, out, sel);
		return r.run();
	} catch (const std::bad_alloc &) {
		return LRZGPU_E_NOMEM;
	} catch (...) {
		return LRZGPU_E_INTERNAL;
	}
}

} // namespace lrzgpu
