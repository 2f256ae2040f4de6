// rzip_resolve_mw.h -- K2 on NW wavefronts (included by rzip_scan.hip, inside namespace lrzgpu).
//
// The one-wavefront resolver of rounds 1 and 2 (tools/experiments/resolver_one_wavefront.hip.inc) is bound by the
// instruction issue of ONE wavefront: a round costs ~5000 instructions
// whatever the number of lanes it commits.  Here the same round runs on NW wavefronts of one
// workgroup (their own SIMDs) over a window of 64 * NW candidates, lane gi = 64 * wave + lane in
// candidate order:
//   * phase A (simulations) and phase D (table writes) are per lane: nothing changes;
//   * phase C's in-order quantities (hash_count prefix, the k-th clean victim, the round-robin
//     eviction index, the first lane that cannot be committed) are prefix counts over ballots: every
//     wave publishes its ballots in LDS and adds the popcounts of the waves before it;
//   * the conflict filter is one LDS table for the workgroup; the exact test of a flagged lane runs
//     over a compacted list of suspects that every wave checks against its own lanes' writes;
//   * the automaton itself (masks, hash_count, sweep pointer, lazy match, records) lives in wave 0,
//     which publishes what the others need at the start of a round and makes every exact serial step;
//   * the window only moves through LDS when a round stops early; a full commit empties it.
// Bit-exact by the same argument as on one wavefront: a prefix of the window is committed only if every lane
// in it was simulated against a table that no earlier lane of the round writes into its read interval.
// Twins are predicted inside a wave (a pair across two waves takes the conflict path).
#pragma once

constexpr int MW_NONE = 0x7FFFFFFF;

struct MwUniform { // published by wave 0 before the first barrier of a round
	u64 min_mask, tag_mask;
	i64 last_match, p_skip, hash_count, clean_ptr, victim_round;
	int wcount_old, topup, ring_head, mode; // mode: 0 batch round, 1 serial steps (shift), 2 exit
	uint32_t abs0;                          // absolute queue index of the first entry taken this round
	int shift;                              // mode 1: lanes consumed by wave 0
	int nv;
	i64 scan_end;
};

template <int NW, int MAXH, int MAXE, bool DENSE = false>
__global__ void __launch_bounds__(64 * NW) k_resolve_mw(const uint8_t *__restrict__ buf, Slot *__restrict__ tbl, ScanState *__restrict__ st, i64 seg_lo,
							int ntiles, const uint32_t *__restrict__ cand_rel, const u64 *__restrict__ cand_tag,
							const uint32_t *__restrict__ tile_count, MatchRec *__restrict__ records, int batch_mode,
							const uint32_t *__restrict__ tile_base, const uint32_t *__restrict__ comp_rel,
							const u64 *__restrict__ comp_tag, uint32_t comp_cap, uint8_t *__restrict__ rank_bytes,
							uint8_t *__restrict__ fp_bytes)
{
	constexpr int W = 64 * NW;
	constexpr int RINGN = NW >= 8 ? 2 * W : 4 * W;
	static_assert(!DENSE || NW == 1, "the dense variant is one wavefront");
	constexpr int PWB = DENSE ? 64 : 16; // suspects per wave that get the exact test
	constexpr int CFB = CF_BITS + (NW >= 4 ? 1 : 0); // four times the writes per round: twice the counters
	constexpr int CFW = (1 << CFB) / 2;
	__shared__ i64 ring_pos[RINGN];
	__shared__ u64 ring_tag[RINGN];
	__shared__ u64 stk_t[64];
	__shared__ i64 stk_off[64];
	__shared__ i64 stk_h[64];
	// phase A scratch of each wave (tag hits of the lookup walk); between rounds the staging area of the window
	__shared__ __attribute__((aligned(16))) i64 hit_all[NW][MAXH * 64];
	__shared__ uint32_t eqs_lds[MAXE * W];
	__shared__ uint32_t cf_bits[CFW];
	__shared__ uint32_t vict[W];
	__shared__ MwUniform U;
	__shared__ u64 xm_x[NW], xm_ev[NW], xm_clean[NW], xm_bad[NW];
	__shared__ int xm_nfl[NW];
	__shared__ int x_why[NW];
	__shared__ i64 x_P[NW];
	__shared__ u64 x_T[NW];
	__shared__ int x_cnt[NW][4];
	__shared__ uint32_t x_lastvict[NW];
	// suspects of the conflict filter: one list per wave (lane order), checked by that wave and the waves before it
	__shared__ uint32_t fl_k[NW][PWB], fl_lo[NW][PWB], fl_hi[NW][PWB];
	__shared__ int fl_res[NW][NW][PWB];
	__shared__ i64 x_miss;
	// ---- the dense variant (see "dense" below) ----
	constexpr int DH = DENSE ? MAXH * 64 : 1, DEM = 16;
	__shared__ uint32_t hit_fr[DH];                          // measured extents of the tag hits, beside hit_all
	__shared__ u64 q_cache[DENSE ? 32 * 64 : 1];             // a walk step's 4th .. 19th fingerprint match (tag, offset), per lane
	// the soft writers of a round by slot, by tag, by (tag, bytes after), by (tag, bytes before): exact-key tables of 128
	// entries (open addressing, at most 64 keys), every entry a mask of the lanes that wrote the key
	constexpr int SPN = DENSE ? 128 : 1;
	__shared__ uint32_t sp_skey[SPN];
	__shared__ u64 sp_smask[SPN], sp_tkey[SPN], sp_tmask[SPN], sp_tnofb[SPN], sp_fkey[SPN], sp_fmask[SPN], sp_bkey[SPN], sp_bmask[SPN];
	__shared__ i64 em_p[DEM], em_ofs[DEM], em_len[DEM];      // matches emitted inside the round, in order
	__shared__ int em_lane[DEM];
	static_assert(sizeof(i64) * MAXH * 64 * NW >= (size_t)W * 128, "staging area");

	const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, gi = threadIdx.x;
	const bool master = wave == 0;
	const u64 lane_bit = 1ull << lane;
	const u64 lanes_below = lane_bit - 1;
	i64 *hit_lds = hit_all[wave];

	Automaton A; // wave 0: the automaton; the others: the read-only part their simulations need
	Resolver &R = A.R;
	R.buf = buf;
	R.tbl = tbl;
	R.rk = rank_bytes;
	R.fpa = fp_bytes;
	R.lane = lane;
	R.hmask = ((u64)1 << st->hash_bits) - 1;
	R.end = st->end;
	R.last_match = st->last_match;
	R.tag_mask = st->tag_mask;
	R.min_mask = st->min_mask;
	R.hash_count = st->hash_count;
	R.hash_limit = st->hash_limit;
	R.clean_ptr = st->clean_ptr;
	R.victim_round = st->victim_round;
	R.max_chain = st->max_chain_len;
	R.tag_hits = st->tag_hits;
	R.tag_misses = st->tag_misses;
	R.stk_t = stk_t;
	R.stk_off = stk_off;
	R.stk_h = stk_h;
	R.hint_p = st->hint_p;
	R.hint_op = st->hint_op;
	R.hint_len = st->hint_len;
	R.allow_abort = true;
	R.aborted = false;
	R.ext_p = R.ext_op = R.ext_done = 0;
	for (int k = threadIdx.x; k < CFW; k += W)
		cf_bits[k] = 0;
	if constexpr (DENSE)
		for (int k = threadIdx.x; k < SPN; k += W) {
			sp_skey[k] = 0xFFFFFFFFu;
			sp_tkey[k] = sp_fkey[k] = sp_bkey[k] = ~0ull;
			sp_smask[k] = sp_tmask[k] = sp_tnofb[k] = sp_fmask[k] = sp_bmask[k] = 0;
		}
	if (threadIdx.x == 0)
		x_miss = 0;

	// ---- wave 0 only ----
	i64 &p_skip = A.p_skip, &cur_p = A.cur_p, &cur_ofs = A.cur_ofs, &cur_len = A.cur_len;
	i64 &n_rec = A.n_rec, &inserts = A.inserts, &lookups = A.lookups, &serial_n = A.serial_n;
	int &error = A.error;
	p_skip = st->p_skip;
	cur_p = st->cur_p;
	cur_ofs = st->cur_ofs;
	cur_len = st->cur_len;
	n_rec = st->n_records;
	A.rec_cap = st->rec_cap;
	inserts = st->inserts;
	lookups = st->lookups;
	error = st->error;
	serial_n = 0; // (dbg[2] at the end)
	A.records = records;
	// round / stop counters and the profile laps of wave 0: in LDS, bumped by one lane (as sixteen 64-bit values per lane
	// they were 32 of the 512 registers of a kernel that spills)
	__shared__ i64 dbg[16];
	if (threadIdx.x < 16)
		dbg[threadIdx.x] = 0;
	auto bump = [&](int slot, i64 by) {
		if (threadIdx.x == 0)
			dbg[slot] += by;
	};
	u64 tclk = __builtin_amdgcn_s_memtime();
	const bool prof = (batch_mode & 2) != 0 && master; // shader-clock laps of wave 0 between the barriers of a round
	auto lap = [&](int slot) {
		if (!prof)
			return;
		const u64 now = __builtin_amdgcn_s_memtime();
		bump(slot, (i64)(now - tclk));
		tclk = now;
	};
	i64 miss_acc = 0; // per lane, every wave
	const i64 tbl_size = (i64)R.hmask + 1;

	// (wave 0) one exact automaton step at candidate (P, T)
	// (inlined at both of its places: as a function of its own it takes the automaton by reference, which puts the
	// automaton into scratch memory for the whole kernel -- 17 % on the rounds)
	auto serial_step = [&](i64 P, u64 T) __attribute__((always_inline)) { A.step(P, T); };

	// ---- candidate queue (LDS), filled by wave 0 from the K1 lists ----
	int tile = 0;
	uint32_t tb0 = 0;
	int ring_head = 0, ring_cnt = 0;
	uint32_t popped_abs = 0;
	const uint32_t ctotal = tile_base[ntiles];
	const bool packed = ctotal <= comp_cap;
	uint32_t cpos = 0;
	i64 skip_seen = p_skip;
	auto push = [&](bool ok, i64 pos, u64 tag) {
		const u64 m = __ballot(ok);
		if (ok) {
			const int slot = (ring_head + ring_cnt + __popcll(m & lanes_below)) & (RINGN - 1);
			ring_pos[slot] = pos;
			ring_tag[slot] = tag;
		}
		ring_cnt += __popcll(m);
	};
	i64 pre_pos[4] = {-1, -1, -1, -1};
	u64 pre_tag[4] = {0, 0, 0, 0};
	uint32_t pre_at = 0xFFFFFFFFu; // packed-list position pre_* were fetched for
	auto fetch4 = [&](uint32_t at, i64(&pos)[4], u64(&tag)[4]) {
#pragma unroll
		for (int j = 0; j < 4; j++) {
			pos[j] = -1;
			tag[j] = 0;
			if (at + 64 * j + lane < ctotal) {
				pos[j] = seg_lo + (i64)comp_rel[at + 64 * j + lane];
				tag[j] = comp_tag[at + 64 * j + lane];
			}
		}
	};
	auto refill_ring = [&]() {
		if (packed) {
			if (p_skip != skip_seen) { // a match was emitted: jump over the candidates inside it
				skip_seen = p_skip;
				if (p_skip >= seg_lo) {
					const i64 t = (p_skip + 1 - seg_lo) / TILE;
					const uint32_t c0 = t < ntiles ? tile_base[t] : ctotal;
					if (c0 > cpos)
						cpos = c0;
				}
			}
			// four independent loads per trip: one memory latency for up to 256 candidates (none when the
			// trip's candidates were fetched during the previous round)
			while (ring_cnt <= RINGN - 256 && cpos < ctotal) {
				i64 pos[4];
				u64 tag[4];
				if (pre_at == cpos) {
#pragma unroll
					for (int j = 0; j < 4; j++) {
						pos[j] = pre_pos[j];
						tag[j] = pre_tag[j];
					}
				} else
					fetch4(cpos, pos, tag);
				pre_at = 0xFFFFFFFFu;
#pragma unroll
				for (int j = 0; j < 4; j++)
					push(pos[j] > p_skip && (tag[j] & R.min_mask) == R.min_mask, pos[j], tag[j]);
				cpos += 256;
			}
			return;
		}
		while (ring_cnt <= RINGN - 64 && tile < ntiles) {
			const uint32_t cnt = tile_count[tile];
			if (tb0 >= cnt || seg_lo + (i64)(tile + 1) * TILE - 1 <= p_skip) { // exhausted / inside a match
				tile++;
				tb0 = 0;
				continue;
			}
			const size_t base = (size_t)tile * TILE;
			i64 pos = -1;
			u64 tag = 0;
			if (tb0 + lane < cnt) {
				pos = seg_lo + (i64)cand_rel[base + tb0 + lane];
				tag = cand_tag[base + tb0 + lane];
			}
			push(pos > p_skip && (tag & R.min_mask) == R.min_mask, pos, tag);
			tb0 += 64;
		}
	};

	// ---- the window: one candidate per lane of the workgroup, gi order = candidate order ----
	int wcount = 0;
	i64 w_pos = -1;
	u64 w_tag = 0;
	bool w_simd = false;
	int w_ticket = 0; // index into eqs_lds, stable while the entry is in the window
	LaneSim L;
	memset(&L, 0, sizeof(L));
	L.k0 = -1;

	// staging layout (SoA over gi, in the hit scratch)
	uint8_t *stage = reinterpret_cast<uint8_t *>(&hit_all[0][0]);
	i64 *sg_pos = reinterpret_cast<i64 *>(stage);
	u64 *sg_tag = reinterpret_cast<u64 *>(stage + (size_t)W * 8);
	u64 *sg_wt = reinterpret_cast<u64 *>(stage + (size_t)W * 16);   // [4][W]
	i64 *sg_woff = reinterpret_cast<i64 *>(stage + (size_t)W * 48); // [4][W]
	uint32_t *sg_u32 = reinterpret_cast<uint32_t *>(stage + (size_t)W * 80); // [10][W]: flags, packed, lo, hi, tw_slot, ticket, w_slot[4]
	// drops the first c entries of the window (c and wcount are the same in every wave)
	auto shift_window = [&](int c) {
		if (c <= 0)
			return;
		if (c >= wcount) {
			wcount = 0;
			w_simd = false;
			return;
		}
		if (gi >= c && gi < wcount) {
			sg_pos[gi] = w_pos;
			sg_tag[gi] = w_tag;
			sg_u32[0 * W + gi] = (uint32_t)w_simd | (uint32_t)L.complex_ << 1 | (uint32_t)L.match << 2 | (uint32_t)L.ins << 3 |
					     (uint32_t)L.victim << 4 | (uint32_t)L.twin << 5 | (uint32_t)L.tw_over << 6 | (uint32_t)(L.dec != 0) << 7 |
					     (uint32_t)(DENSE && L.soft1) << 8;
			sg_u32[1 * W + gi] = (uint32_t)(L.tw_kind & 3) | (uint32_t)((L.k0 + 1) & 7) << 2 | (uint32_t)(L.nw & 7) << 5 | (uint32_t)(L.misses & 0xFFFF) << 8;
			sg_u32[2 * W + gi] = L.lo;
			sg_u32[3 * W + gi] = L.hi;
			sg_u32[4 * W + gi] = L.tw_slot;
			sg_u32[5 * W + gi] = (uint32_t)w_ticket;
#pragma unroll
			for (int k = 0; k < 4; k++) {
				sg_u32[(6 + k) * W + gi] = L.w_slot[k];
				sg_wt[k * W + gi] = L.w_t[k];
				sg_woff[k * W + gi] = L.w_off[k];
			}
		}
		__syncthreads();
		const int src = gi + c;
		if (src < wcount) {
			w_pos = sg_pos[src];
			w_tag = sg_tag[src];
			const uint32_t fl = sg_u32[0 * W + src], pk = sg_u32[1 * W + src];
			w_simd = (fl & 1) != 0;
			L.complex_ = (fl & 2) != 0;
			L.match = (fl & 4) != 0;
			L.ins = (fl & 8) != 0;
			L.victim = (fl & 16) != 0;
			L.twin = (fl & 32) != 0;
			L.tw_over = (fl & 64) != 0;
			L.dec = (fl >> 7) & 1;
			L.soft1 = (fl >> 8) & 1;
			L.tw_kind = (int)(pk & 3);
			L.k0 = (int)((pk >> 2) & 7) - 1;
			L.nw = (int)((pk >> 5) & 7);
			L.misses = (int)(pk >> 8);
			if constexpr (DENSE) {
				L.pot = L.match; // (a kept lane never has possible matches: those are simulated again, their hits are gone)
				L.nh = 0;
			}
			L.lo = sg_u32[2 * W + src];
			L.hi = sg_u32[3 * W + src];
			L.tw_slot = sg_u32[4 * W + src];
			w_ticket = (int)sg_u32[5 * W + src];
#pragma unroll
			for (int k = 0; k < 4; k++) {
				L.w_slot[k] = sg_u32[(6 + k) * W + src];
				L.w_t[k] = sg_wt[k * W + src];
				L.w_off[k] = sg_woff[k * W + src];
			}
		}
		wcount -= c;
		if (gi >= wcount)
			w_simd = false;
		// (the staging area is the hit scratch of phase A: the first barrier of the next round lies in between)
	};

	// When rounds stop paying the automaton takes a stretch of exact steps and then tries rounds again.  A round is a
	// poor one when it stopped early having committed fewer than POOR_COMMIT candidates: a round costs 12 us on text and
	// 35 us where every probe window is full, an exact step 2 to 3 us, so below eight candidates the steps are the cheaper
	// way.  The rule counts candidates and nothing else: rounds 1 to 5 judged a round by the shader clock against an
	// assumed cost of a step, which made a branch of this kernel depend on what s_memtime counts on the box (on one whose
	// counter ran slower than assumed no round was ever poor).  Poor rounds raise a score, good ones lower it
	// twice as fast; at POOR_SCORE a stretch begins, and every stretch that ends where it began -- two more poor rounds --
	// is twice as long as the one before, until the score is back at zero.  (With fixed stretches of 256 steps a 5 MiB
	// input of four symbols spent 3 of its 8 s in the rounds in between, one of random phrases a round per match:
	// profiles/r5_serial_step.log.)  batch_mode bits 2 and 3 are the test hooks that force the verdict: no round is ever
	// poor / every round is (LRZGPU_RESOLVE_POOR=never|always, read per scan): both paths are parity-tested on every box.
	constexpr int SPAN_MIN = 256, SPAN_MAX = 16384, POOR_SCORE = 8, POOR_COMMIT = 8;
	int poor_rounds = 0, serial_left = 0, serial_span = SPAN_MIN;
	// the two variants hand over to each other through the host (scan_chunk_device): this one asks for the dense variant
	// when its rounds stop paying (leave = 4), the dense variant gives back when fewer than one in eight of DENSE_WINDOW rounds had a
	// use for what it adds (leave = 5); the kernel ends with ScanState::error = leave and p_skip in front of the first
	// candidate it has not examined
	constexpr int DENSE_WINDOW = 256;
	int leave = 0, dense_rounds = 0, dense_used = 0;
	__syncthreads();
	for (;;) {
		// ---- wave 0: queue, top-up size, state for the others ----
		if (master) {
			int mode = 0, k = 0;
			if (leave && !error && cur_len == 0) {
				// (between two matches only: the next launch starts a new segment at p_skip + 1)
				refill_ring();
				i64 next_p = -1;
				if (wcount > 0)
					next_p = (i64)readlane64((u64)w_pos, 0);
				else if (ring_cnt > 0)
					next_p = ring_pos[ring_head];
				if (next_p >= 0) {
					if (next_p - 1 > p_skip)
						p_skip = next_p - 1;
					error = leave;
				}
				leave = 0;
			}
			if (error)
				mode = 2;
			else {
				refill_ring();
				k = W - wcount;
				if (k > ring_cnt)
					k = ring_cnt;
				if (wcount + k == 0)
					mode = 2;
				else if (!(batch_mode & 1) || (!DENSE && cur_len > 0) || serial_left > 0)
					mode = 1; // (dense: a match in the making is carried through the rounds)
			}
			if (lane == 0) {
				U.min_mask = R.min_mask;
				U.tag_mask = R.tag_mask;
				U.last_match = R.last_match;
				U.p_skip = p_skip;
				U.hash_count = R.hash_count;
				U.clean_ptr = R.clean_ptr;
				U.victim_round = R.victim_round;
				U.wcount_old = wcount;
				U.topup = k;
				U.ring_head = ring_head;
				U.abs0 = popped_abs;
				U.mode = mode;
			}
			ring_head = (ring_head + k) & (RINGN - 1);
			ring_cnt -= k;
			popped_abs += (uint32_t)k;
		}
		lap(8); // queue refill, publication
		__syncthreads(); // #1
		const int mode = U.mode;
		if (mode == 2)
			break;
		if (!master) {
			R.min_mask = U.min_mask;
			R.tag_mask = U.tag_mask;
			R.last_match = U.last_match;
			p_skip = U.p_skip;
			R.hash_count = U.hash_count;
			R.clean_ptr = U.clean_ptr;
			R.victim_round = U.victim_round;
		}
		{
			const int k = U.topup, rh = U.ring_head;
			if (gi >= wcount && gi < wcount + k) {
				const int slot = (rh + gi - wcount) & (RINGN - 1);
				w_pos = ring_pos[slot];
				w_tag = ring_tag[slot];
				w_simd = false;
				w_ticket = (int)((U.abs0 + (uint32_t)(gi - wcount)) & (W - 1));
			}
			wcount += k;
		}
		const bool has = gi < wcount;
		const bool alive = has && w_pos > p_skip && (w_tag & R.min_mask) == R.min_mask;
		// wave 0: what the next refill and this round's sweep will read first, fetched behind the simulations
		u64 pre_rb = 0;
		if (master && mode == 0) {
			if (packed && pre_at != cpos && cpos < ctotal) {
				fetch4(cpos, pre_pos, pre_tag);
				pre_at = cpos;
			}
			if (R.hash_count + W > R.hash_limit && R.clean_ptr + 8 * lane < tbl_size)
				pre_rb = reinterpret_cast<const U64u *>(R.rk + R.clean_ptr + 8 * lane)->v;
		}

		// ---- serial path: pending lazy match, batching disabled, or batching not paying off ----
		if (mode == 1) {
			if (master) {
				const int lim = wcount < 64 ? wcount : 64;
				int c = 0;
				while (c < lim && !error && (c == 0 || !(batch_mode & 1) || (!DENSE && cur_len > 0) || serial_left > 0)) {
					if (serial_left > 0)
						serial_left--;
					const i64 P = (i64)readlane64((u64)w_pos, c);
					const u64 T = readlane64(w_tag, c);
					if (P > p_skip && (T & R.min_mask) == R.min_mask)
						serial_step(P, T);
					c++;
				}
				if (lane == 0)
					U.shift = c;
			}
			__syncthreads();
			w_simd = false; // the table changed in ways the simulations did not see
			shift_window(U.shift);
			continue;
		}

		const u64 better = mask_up(R.min_mask);
		const i64 hc0 = R.hash_count, cp0 = R.clean_ptr, vr0 = R.victim_round;
		const u64 tm0 = R.tag_mask;

		// ---- phase A/B: simulations, every wave for its own lanes ----
		{
			const bool need_sim = alive && !w_simd;
			if (__ballot(need_sim)) {
				simulate_lanes<MAXH, MAXE, DENSE>(R, tbl, buf, tbl_size, better, lane, alive, need_sim, w_tag, w_pos, w_ticket, hit_lds, eqs_lds, W, L, [](int) {}, hit_fr, q_cache);
				if (need_sim)
					w_simd = true;
			}
		}

		// ---- dense: the lazy-match automaton over the window, in candidate order (src/rzip.c:697-731) ----
		// In the dense variant a candidate whose lookup finds a match does not end the round.  Every lane has measured
		// its tag hits (simulate_lanes, measure_hit); here the wavefront replays what hash_search does with them one
		// candidate after the other: the longer match is kept (first of equal lengths: an inclusive prefix maximum over
		// (length, -lane)), a match is emitted at the first candidate MINIMUM_MATCH behind its start, the candidates
		// inside it are dead, the ones behind it price their hits again under the new last_match (which clips backward
		// extents) -- until nothing more is emitted or a candidate needs the exact step (d_stop).  Nothing is written
		// here: the emissions go to LDS and count only as far as the round commits (phase D).
		if constexpr (DENSE)
			lap(9); // simulations (the in-order pass below: slot 4, which counts nothing in this variant)
		bool live_d = alive;
		int d_stop = MW_NONE, n_em = 0;
		i64 post_len = 0, post_p = 0, post_ofs = 0, post_lm = 0; // the match in the making and last_match AFTER this lane
		int ev_real = 0, ev_miss = 0;                             // what a lane's hits are worth under the last_match it meets
		u64 d_fa = 0, d_ba = 0;                                   // the 8 bytes after / before the candidate (pair test below)
		bool d_fb_ok = false;
		if constexpr (DENSE) {
			if (alive && w_pos >= 8 && R.end - w_pos >= 8) {
				d_fa = reinterpret_cast<const U64u *>(buf + w_pos)->v;
				d_ba = reinterpret_cast<const U64u *>(buf + w_pos - 8)->v;
				d_fb_ok = true;
			}
			i64 c_len = cur_len, c_p = cur_p, c_ofs = cur_ofs, lm = R.last_match; // (wave-uniform)
			int start = 0;
			for (;;) {
				i64 b_len = 0, b_off = 0, b_rev = 0;
				bool unk = false;
				const bool mine = live_d && gi >= start;
				if (mine && L.pot && !L.complex_) {
					const i64 mb = w_pos - (lm > 0 ? lm : 0);
					int real = 0, miss = 0;
					for (int k = 0; k < L.nh; k++) {
						const uint32_t fr = hit_fr[k * 64 + lane];
						const i64 fwd = (i64)(fr & 0xFFFFu), back = (i64)((fr >> 16) & 0x7FFFu);
						const i64 rev = back < mb ? back : mb;
						if ((fr >> 31) && mb > back)
							unk = true; // the backward compare stopped at its cap and last_match allows more
						const i64 len = fwd + rev;
						if (len < MINIMUM_MATCH)
							miss++;
						else {
							real++;
							if (len > b_len) {
								b_len = len;
								b_rev = rev;
								b_off = hit_lds[k * 64 + lane] - rev;
							}
						}
					}
					ev_real = real;
					ev_miss = miss + (L.twin ? 1 : 0);
				}
				const u64 barm = __ballot(mine && (L.complex_ || unk));
				const int barrier = barm ? __ffsll((long long)barm) - 1 : 64;
				const bool inr = gi >= start && gi < barrier;
				uint32_t key = (mine && inr) ? ((uint32_t)b_len << 6 | (uint32_t)(63 - lane)) : 0u;
#pragma unroll
				for (int d = 1; d < 64; d <<= 1) {
					const uint32_t o = __shfl_up(key, d);
					if (lane >= d && o > key)
						key = o;
				}
				const int src = 63 - (int)(key & 63u);
				const i64 run_len = (i64)(key >> 6);
				const i64 s_rev = (i64)bcast64((u64)b_rev, src), s_off = (i64)bcast64((u64)b_off, src), s_pos = (i64)bcast64((u64)w_pos, src);
				i64 my_len = c_len, my_p = c_p, my_ofs = c_ofs;
				if (run_len > c_len) {
					my_len = run_len;
					my_p = s_pos - s_rev;
					my_ofs = s_off;
				}
				const bool emit = mine && inr && my_len >= MINIMUM_MATCH && (my_len >= GREAT_MATCH || w_pos >= my_p + MINIMUM_MATCH);
				const u64 em = __ballot(emit);
				if (inr) {
					post_len = my_len;
					post_p = my_p;
					post_ofs = my_ofs;
					post_lm = lm;
				}
				if (!em) {
					if (barrier < 64)
						d_stop = barrier;
					break;
				}
				const int e = __ffsll((long long)em) - 1;
				const i64 E_len = (i64)readlane64((u64)my_len, e), E_p = (i64)readlane64((u64)my_p, e), E_ofs = (i64)readlane64((u64)my_ofs, e);
				const i64 E_P = (i64)readlane64((u64)w_pos, e);
				const i64 nlm = E_p + E_len;
				if (E_P > nlm || n_em >= DEM || n_rec + n_em >= A.rec_cap) {
					d_stop = e; // the candidate stands again behind its own emission (or no room): the exact step
					break;
				}
				if (lane == 0) {
					em_lane[n_em] = e;
					em_p[n_em] = E_p;
					em_ofs[n_em] = E_ofs;
					em_len[n_em] = E_len;
				}
				n_em++;
				if (gi == e) {
					post_len = 0;
					post_p = nlm;
					post_lm = nlm;
				}
				c_len = 0;
				c_p = nlm;
				c_ofs = E_ofs;
				lm = nlm;
				if (gi > e && w_pos <= nlm)
					live_d = false;
				start = e + 1;
			}
		}

		if constexpr (DENSE)
			lap(4);
		// ---- phase C: the longest conflict-free prefix of the window ----
		const bool live = DENSE ? live_d : alive;
		const bool cut_match = !DENSE && L.match; // (the dense variant carries matches through the round)
		const int x = (live && L.ins && !L.dec && !L.complex_ && !cut_match) ? 1 : 0;
		const bool evicts = live && L.victim && !L.complex_ && !cut_match;
		// dense: my one write puts an entry of my tag where one of my tag was (round-robin eviction, or kind 1 over my own tag)
		const bool soft_w = DENSE && (evicts || (live && L.soft1 && L.nw == 1 && !L.complex_));
		{
			const u64 bx = __ballot(x != 0), be = __ballot(evicts);
			if (lane == 0) {
				xm_x[wave] = bx;
				xm_ev[wave] = be;
			}
		}
		__syncthreads(); // #2
		lap(9); // top-up + simulations (the slowest wave)
		int x_before = 0, ev_before = 0;
		u64 evict_all = 0;
#pragma unroll
		for (int w2 = 0; w2 < NW; w2++) {
			if (w2 < wave) {
				x_before += __popcll(xm_x[w2]);
				ev_before += __popcll(xm_ev[w2]);
			}
			evict_all |= xm_ev[w2];
		}
		const int px = x + x_before + __popcll(xm_x[wave] & lanes_below); // inclusive prefix count
		const i64 hc_before = hc0 + (px - x) < R.hash_limit ? hc0 + (px - x) : R.hash_limit;
		const bool cleans = x && hc_before + 1 > R.hash_limit;
		if (evicts) {
			const uint32_t r = (uint32_t)((vr0 + ev_before + __popcll(xm_ev[wave] & lanes_below)) % (i64)R.max_chain);
			L.w_slot[0] = eqs_lds[r * W + w_ticket];
		}
		(void)evict_all;
		{
			const u64 bc = __ballot(cleans);
			if (lane == 0)
				xm_clean[wave] = bc;
		}
		__syncthreads(); // #3
		int kth = __popcll(xm_clean[wave] & lanes_below), want = 0, first_clean_gi = MW_NONE;
#pragma unroll
		for (int w2 = 0; w2 < NW; w2++) {
			const u64 m = xm_clean[w2];
			if (w2 < wave)
				kth += __popcll(m);
			want += __popcll(m);
			if (m && first_clean_gi == MW_NONE)
				first_clean_gi = 64 * w2 + __ffsll((long long)m) - 1;
		}
		// victim list: the next `want` entries the sweep would delete, in sweep order (wave 0)
		const int vic_nb1 = __popcll(better) + 1;
		int nv = 0;
		i64 scan_end = cp0;
		if (want) {
			if (master) {
				// 512 slots per trip: eight rank bytes per lane, in slot order (lane-major), one memory latency
				i64 ptr = cp0;
				int n = 0;
				const u64 nb8 = (u64)vic_nb1 * B01;
				while (n < want && ptr < tbl_size && ptr - cp0 < (i64)32768 * NW) {
					const i64 q0 = ptr + 8 * lane;
					u64 rb = 0;
					if (ptr == cp0 && hc0 + W > R.hash_limit)
						rb = pre_rb; // (nothing has written the table since)
					else if (q0 < tbl_size)
						rb = reinterpret_cast<const U64u *>(R.rk + q0)->v; // (the array is zero-padded past the table)
					uint32_t m8 = flags_to_bits(bytes_lt(rb, nb8) & ~bytes_lt(rb, B01)); // occupied and due for cleaning
					if (q0 + 8 > tbl_size)
						m8 &= q0 < tbl_size ? (1u << (int)(tbl_size - q0)) - 1 : 0;
					int before = 0, total = 0;
#pragma unroll
					for (int b = 0; b < 8; b++) {
						const u64 bm = __ballot((m8 >> b) & 1);
						before += __popcll(bm & lanes_below);
						total += __popcll(bm);
					}
					int r = n + before;
					while (m8) {
						const int b = __ffs((int)m8) - 1;
						m8 &= m8 - 1;
						if (r < W)
							vict[r] = (uint32_t)(q0 + b);
						r++;
					}
					n += total;
					ptr += 512;
				}
				if (lane == 0) {
					U.nv = n > W ? W : n;
					U.scan_end = ptr < tbl_size ? ptr : tbl_size;
				}
			}
			__syncthreads(); // #4 (want is the same in every wave)
			nv = U.nv;
			scan_end = U.scan_end;
		}
		lap(10); // prefix counts, victim sweep
		uint32_t my_vict = 0xFFFFFFFFu;
		int why = 0; // 3 complex, 4 match, 5 conflict, 6 no victim, 7 swept range
		bool stop = false;
		if (live && L.complex_) {
			stop = true;
			why = 3;
		} else if (live && cut_match) {
			stop = true;
			why = 4;
		}
		if (DENSE && live && gi == d_stop && !stop) {
			stop = true;
			why = 3;
		}
		if (cleans) {
			if (kth < nv)
				my_vict = vict[kth];
			else if (!stop) {
				stop = true; // sweep wrap / no victim in reach: serial path
				why = 6;
			}
		}
		// the first clean of a chunk switches tag_mask from the initial mask to `better`
		if (tm0 != better && first_clean_gi != MW_NONE && live && !stop && gi > first_clean_gi) {
			stop = true;
			why = 3;
		}
		// an insert landing inside the swept range could change what the sweep meets
		if (live && want && !stop)
			for (int k = 0; k < 4; k++)
				if (k < L.nw && (i64)L.w_slot[k] >= cp0 && (i64)L.w_slot[k] < scan_end) {
					stop = true;
					why = 7;
				}
		// twin successors: did the predecessor's simulation make exactly the predicted insert?
		const bool tw_live = live && L.twin && !L.complex_ && !cut_match;
		if (__ballot(tw_live)) {
			const uint32_t p_slot = __shfl_up(L.w_slot[0], 1);
			const int p_nw = __shfl_up(L.nw, 1), p_k0 = __shfl_up(L.k0, 1);
			const int p_ok = __shfl_up((int)(live && L.ins && !L.complex_ && !cut_match && !L.victim && !stop), 1);
			if (tw_live && !stop && !(lane > 0 && p_ok && p_nw >= 1 && p_slot == L.tw_slot && p_k0 == L.tw_kind)) {
				stop = true;
				why = 5;
			}
		}
		// conflicts: the EARLIEST lane whose write lies inside my read interval (see k_resolve)
		int gsh = __popcll(R.min_mask) - 2;
		gsh = gsh < 3 ? 3 : gsh > 16 ? 16 : gsh;
		uint32_t wr[5], wh[5];
#pragma unroll
		for (int k = 0; k < 5; k++) {
			wr[k] = 0xFFFFFFFFu;
			if (live) {
				if (k < 4) {
					if (k < L.nw)
						wr[k] = L.w_slot[k];
				} else
					wr[k] = my_vict;
			}
			// (dense: a write that puts my tag where my tag was is no write for the filter and the exact test -- the pass
			// over the soft writers below deals with it)
			wh[k] = wr[k] != 0xFFFFFFFFu && !(k == 0 && soft_w) ? ((wr[k] >> gsh) * 2654435761u) >> (32 - CFB) : 0xFFFFFFFFu;
		}
#pragma unroll
		for (int k = 0; k < 5; k++)
			if (wh[k] != 0xFFFFFFFFu)
				__hip_atomic_fetch_add(&cf_bits[wh[k] >> 1], 1u << (16 * (wh[k] & 1)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
		__syncthreads(); // #5
		lap(14); // stops, twin check, filter writes
		const bool reads = live && !L.complex_ && !cut_match;
		const uint32_t r_lo = L.lo & ~7u, r_hi = L.hi | 7u;
		const uint32_t tw_h = tw_live && !stop ? ((L.tw_slot >> gsh) * 2654435761u) >> (32 - CFB) : 0xFFFFFFFFu;
		bool flagged = false;
		// the writes of the lane after me come after me whatever they are: not counted either (a twin's
		// insert lands in its predecessor's granule, which made a suspect of every predecessor)
		uint32_t nh[5];
#pragma unroll
		for (int q = 0; q < 5; q++) {
			nh[q] = __shfl_down(wh[q], 1);
			if (lane == 63)
				nh[q] = 0xFFFFFFFFu;
		}
		if (reads) {
			const uint32_t g1 = L.hi >> gsh;
			uint32_t expect = tw_h != 0xFFFFFFFFu ? 1u : 0u;
			for (uint32_t g = L.lo >> gsh; g <= g1; g++) {
				const uint32_t hb = (g * 2654435761u) >> (32 - CFB);
				uint32_t cnt = (cf_bits[hb >> 1] >> (16 * (hb & 1))) & 0xFFFFu;
#pragma unroll
				for (int q = 0; q < 5; q++)
					cnt -= (uint32_t)(wh[q] == hb) + (uint32_t)(nh[q] == hb);
				if (expect && hb == tw_h && cnt) {
					cnt--;
					expect = 0;
				}
				if (cnt)
					flagged = true;
			}
		}
		lap(15); // filter reads
		int first_conf = MW_NONE; // window index of the earliest conflicting writer
		int sup_by = MW_NONE;     // dense: the later candidate that evicts the slot of my own eviction again
		bool soft_me = false;     // dense: a same-tag eviction in front of me passed as harmless (my kept simulation is stale once it lands)
		int fidx;
		{
			const u64 bf = __ballot(flagged);
			fidx = __popcll(bf & lanes_below);
			if (flagged) {
				if (fidx < PWB) {
					fl_k[wave][fidx] = (uint32_t)gi | (tw_live ? 0x80000000u : 0u); // (a twin's predecessor's insert is part of its simulation)
					fl_lo[wave][fidx] = r_lo;
					fl_hi[wave][fidx] = r_hi;
				} else
					first_conf = 0; // too many suspects: call the rest conflicting (they re-simulate)
			}
			if (lane == 0)
				xm_nfl[wave] = __popcll(bf) < PWB ? __popcll(bf) : PWB;
		}
		__syncthreads(); // #6
		lap(7);
		{
			int any = 0;
#pragma unroll
			for (int w2 = 0; w2 < NW; w2++)
				any += xm_nfl[w2];
			if (any) { // (the same in every wave)
#pragma unroll
				for (int w2 = 0; w2 < NW; w2++) {
					const int n2 = xm_nfl[w2];
					if (w2 < wave || n2 == 0) // writers of a later wave come after every suspect of this list
						continue;
					// the whole list in registers (lane e holds entry e), then one uniform pass per entry
					const uint32_t mk = fl_k[w2][lane & (PWB - 1)], mlo = fl_lo[w2][lane & (PWB - 1)], mhi = fl_hi[w2][lane & (PWB - 1)];
					int res = MW_NONE;
					for (int e = 0; e < n2; e++) {
						const uint32_t kk = (uint32_t)__builtin_amdgcn_readlane((int)mk, e);
						const uint32_t klo = (uint32_t)__builtin_amdgcn_readlane((int)mlo, e), khi = (uint32_t)__builtin_amdgcn_readlane((int)mhi, e);
						const int k = (int)(kk & 0xFFFFu);
						const bool excl0 = (kk >> 31) != 0 && gi == k - 1; // the twin's predecessor: its insert is expected
						const uint32_t span = khi - klo;                 // (an unused write slot, 0xFFFFFFFF, is never inside)
						bool hit = wr[0] - klo <= span && !excl0 && !soft_w;
#pragma unroll
						for (int q = 1; q < 5; q++)
							hit |= wr[q] - klo <= span;
						const u64 hm = __ballot(hit && gi < k);
						if (lane == e && hm)
							res = 64 * wave + __ffsll((long long)hm) - 1;
					}
					if (lane < n2)
						fl_res[wave][w2][lane] = res;
				}
				__syncthreads(); // #8
				if (flagged && fidx < PWB) {
#pragma unroll
					for (int w1 = 0; w1 < NW; w1++)
						if (w1 <= wave) {
							const int r = fl_res[w1][wave][fidx];
							if (r < first_conf)
								first_conf = r;
						}
				}
			}
		}
		if constexpr (DENSE) {
			// Soft writes: a round-robin eviction, or a kind-1 insert over an entry of its own tag, replaces one entry of my
			// tag by another entry of my tag -- the slot's rank byte, fingerprint byte and tag word stay what they are.
			// Only two kinds of later candidates can tell: one that writes that very slot (the later store must be the
			// one that stands), and one of the SAME tag, whose lookup meets my position where the replaced entry was.
			// For a same-tag candidate all of whose hits were misses the exchange is one certain miss for another when
			// its bytes and mine differ within 8 both ways (fwd < 8, back < 8: single_match_len() = 0 under any
			// last_match); anything else is a conflict like a hard write.  One pass over the soft writers, every later
			// lane against the writer's tag, slot and bytes -- no interval, no filter: a lookup always covers its own chain.
			// (32-bit folds stand for the 64-bit byte strings: different folds prove different values, equal folds are taken
			// for equal values -- the cautious side of each test; multiplicative folds: the bytes of a four-letter input
			// XOR-fold onto a few hundred values.)  Rounds 6a walked the soft writers one by one, every later lane against
			// each (64 iterations of ~90 instructions on four-letter data: a third of the round); now every soft writer
			// enters four exact-key tables in LDS -- its slot, its tag, (tag, bytes after), (tag, bytes before) -- with
			// its lane bit, and every lane looks its own keys up: the masks in front of its own bit are the writers it
			// has to answer to, the first bit behind it in its slot's mask is the store that stands instead of its own.
			const uint32_t th = (uint32_t)(w_tag ^ (w_tag >> 32));
			const uint32_t fah = (uint32_t)((d_fa * 0x9E3779B97F4A7C15ull) >> 32), bah = (uint32_t)((d_ba * 0x9E3779B97F4A7C15ull) >> 32);
			const u64 kf = (((u64)th << 32) | fah) & 0x7FFFFFFFFFFFFFFFull, kb = (((u64)th << 32) | bah) & 0x7FFFFFFFFFFFFFFFull;
			const bool passable = L.nw <= 1 && !cleans && !L.pot && d_fb_ok && live && !L.complex_;
			const bool multi = live && (wr[1] != 0xFFFFFFFFu || wr[4] != 0xFFFFFFFFu); // (more write slots than wr[0])
			const bool looks = live && !L.complex_;
			auto ins32 = [&](uint32_t key) -> int {
				uint32_t h = (key * 2654435761u) >> 25;
				for (;;) {
					const uint32_t old = atomicCAS(&sp_skey[h], 0xFFFFFFFFu, key);
					if (old == 0xFFFFFFFFu || old == key)
						return (int)h;
					h = (h + 1) & (SPN - 1);
				}
			};
			auto find32 = [&](uint32_t key) -> int {
				uint32_t h = (key * 2654435761u) >> 25;
				for (;;) {
					const uint32_t v = sp_skey[h];
					if (v == key)
						return (int)h;
					if (v == 0xFFFFFFFFu)
						return -1;
					h = (h + 1) & (SPN - 1);
				}
			};
			auto ins64 = [&](u64 *keys, u64 key) -> int {
				uint32_t h = (uint32_t)((key * 0x9E3779B97F4A7C15ull) >> 57);
				for (;;) {
					const u64 old = atomicCAS((unsigned long long *)&keys[h], ~0ull, (unsigned long long)key);
					if (old == ~0ull || old == key)
						return (int)h;
					h = (h + 1) & (SPN - 1);
				}
			};
			auto find64 = [&](const u64 *keys, u64 key) -> int {
				uint32_t h = (uint32_t)((key * 0x9E3779B97F4A7C15ull) >> 57);
				for (;;) {
					const u64 v = keys[h];
					if (v == key)
						return (int)h;
					if (v == ~0ull)
						return -1;
					h = (h + 1) & (SPN - 1);
				}
			};
			int hs = -1, ht = -1, hf = -1, hb = -1;
			if (__ballot(soft_w)) {
				if (soft_w) {
					hs = ins32(wr[0]);
					atomicOr((unsigned long long *)&sp_smask[hs], (unsigned long long)lane_bit);
					ht = ins64(sp_tkey, w_tag);
					atomicOr((unsigned long long *)&sp_tmask[ht], (unsigned long long)lane_bit);
					if (!d_fb_ok)
						atomicOr((unsigned long long *)&sp_tnofb[ht], (unsigned long long)lane_bit);
					hf = ins64(sp_fkey, kf);
					atomicOr((unsigned long long *)&sp_fmask[hf], (unsigned long long)lane_bit);
					hb = ins64(sp_bkey, kb);
					atomicOr((unsigned long long *)&sp_bmask[hb], (unsigned long long)lane_bit);
				}
				__syncthreads(); // (one wavefront: its LDS operations are in order; this is the compiler's fence)
				u64 conf_m = 0;
				if (looks) {
					u64 cls = 0;
					const int t = find64(sp_tkey, w_tag);
					if (t >= 0)
						cls = sp_tmask[t];
					const u64 ecls = cls & lanes_below; // soft writers of my tag in front of me
					if (ecls) {
						if (!passable)
							conf_m |= ecls;
						else {
							u64 d = sp_tnofb[t];
							const int f2 = find64(sp_fkey, kf), b2 = find64(sp_bkey, kb);
							if (f2 >= 0)
								d |= sp_fmask[f2];
							if (b2 >= 0)
								d |= sp_bmask[b2];
							d &= lanes_below;
							if (d)
								conf_m |= d; // one of them has my bytes within 8, or none to show
							else
								soft_me = true;
						}
					}
#pragma unroll
					for (int q = 0; q < 5; q++) {
						if (wr[q] == 0xFFFFFFFFu || (q > 0 && !multi))
							continue;
						const int s2 = find32(wr[q]);
						if (s2 < 0)
							continue;
						const u64 m = sp_smask[s2];
						// a soft writer of another tag in front of me on a slot I write (of any tag, if I write several)
						conf_m |= m & lanes_below & (multi ? ~0ull : ~cls);
						if (q == 0 && soft_w) {
							const u64 later = m & ~lanes_below & ~lane_bit;
							if (later)
								sup_by = __ffsll((long long)later) - 1; // the next store into my slot: it stands if it commits
						}
					}
				}
				if (conf_m && __ffsll((long long)conf_m) - 1 < first_conf)
					first_conf = __ffsll((long long)conf_m) - 1;
				__syncthreads();
				if (soft_w) { // the tables go back empty (several lanes may clear one entry)
					sp_skey[hs] = 0xFFFFFFFFu;
					sp_smask[hs] = 0;
					sp_tkey[ht] = ~0ull;
					sp_tmask[ht] = 0;
					sp_tnofb[ht] = 0;
					sp_fkey[hf] = ~0ull;
					sp_fmask[hf] = 0;
					sp_bkey[hb] = ~0ull;
					sp_bmask[hb] = 0;
				}
			}
		}
#pragma unroll
		for (int k = 0; k < 5; k++)
			if (wh[k] != 0xFFFFFFFFu)
				cf_bits[wh[k] >> 1] = 0;
		const bool conflict = first_conf != MW_NONE;
		if (!stop && conflict)
			why = 5;
		{
			const u64 bb = __ballot(live && (stop || conflict));
			if (lane == 0)
				xm_bad[wave] = bb;
			if (bb && lane == __ffsll((long long)bb) - 1) {
				x_why[wave] = why;
				x_P[wave] = w_pos;
				x_T[wave] = w_tag;
			}
		}
		__syncthreads(); // #9
		lap(11); // filter reads, suspects
		int f = wcount, why_f = 0;
		i64 P_f = 0;
		u64 T_f = 0;
#pragma unroll
		for (int w2 = NW - 1; w2 >= 0; w2--) {
			const u64 m = xm_bad[w2];
			if (m) {
				f = 64 * w2 + __ffsll((long long)m) - 1;
				why_f = x_why[w2];
				P_f = x_P[w2];
				T_f = x_T[w2];
			}
		}
		if (f > wcount)
			f = wcount;
		const bool committed = gi < f && live;

		// ---- phase D: apply the committed prefix ----
		const bool nxt_over = __shfl_down((int)(gi < f && live && L.twin && L.tw_over), 1) != 0 && lane < 63;
		int n_real = 0; // dense: real matches met by the committed lookups (tag_hits)
		if (committed) {
			const bool superseded = DENSE && soft_w && sup_by < f; // (a later committed store of my tag into the same slot stands)
			for (int k = 0; k < 4; k++)
				if (k < L.nw && !(k == 0 && (nxt_over || superseded)))
					R.store_slot(L.w_slot[k], L.w_t[k], L.w_off[k]);
			if (cleans)
				R.store_slot(my_vict, 0, 0);
			miss_acc += (DENSE && L.pot) ? ev_miss : L.misses;
			if (DENSE && L.pot)
				n_real = ev_real;
		}
		if constexpr (DENSE) {
#pragma unroll
			for (int d = 32; d >= 1; d >>= 1)
				n_real += __shfl_xor(n_real, d);
		}
		{
			const u64 cm = __ballot(committed);
			const int n_ins = __popcll(__ballot(committed && L.ins));
			const int n_x = __popcll(__ballot(committed && x));
			const int n_ev = __popcll(__ballot(committed && evicts));
			const u64 cc = __ballot(committed && cleans);
			const uint32_t lastv = cc ? (uint32_t)__shfl((int)my_vict, 63 - __clzll((long long)cc)) : 0xFFFFFFFFu;
			if (lane == 0) {
				x_cnt[wave][0] = __popcll(cm);
				x_cnt[wave][1] = n_ins;
				x_cnt[wave][2] = n_x;
				x_cnt[wave][3] = n_ev;
				x_lastvict[wave] = lastv;
			}
		}
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
		__syncthreads(); // #10: every table write of the round is visible to every wave
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
		lap(12); // table writes
		uint32_t last_clean = 0xFFFFFFFFu;
#pragma unroll
		for (int w2 = 0; w2 < NW; w2++)
			if (x_lastvict[w2] != 0xFFFFFFFFu)
				last_clean = x_lastvict[w2];
		const bool first_clean = last_clean != 0xFFFFFFFFu && tm0 != better;
		if (master) {
			int n_commit = 0, n_ins = 0, n_x = 0, n_ev = 0;
#pragma unroll
			for (int w2 = 0; w2 < NW; w2++) {
				n_commit += x_cnt[w2][0];
				n_ins += x_cnt[w2][1];
				n_x += x_cnt[w2][2];
				n_ev += x_cnt[w2][3];
			}
			lookups += n_commit;
			bump(0, 1);
			bump(1, n_commit);
			if constexpr (DENSE) {
				// the matches emitted in front of lane f, and the automaton as lane f - 1 left it
				R.tag_hits += n_real;
				int applied = 0;
				for (int q = 0; q < n_em; q++)
					if (em_lane[q] < f) {
						if (lane == 0) {
							MatchRec r;
							r.p = em_p[q];
							r.ofs = em_ofs[q];
							r.len = em_len[q];
							A.records[n_rec] = r;
						}
						n_rec++;
						applied++;
					}
				if (f > 0) {
					cur_len = (i64)readlane64((u64)post_len, f - 1);
					cur_p = (i64)readlane64((u64)post_p, f - 1);
					cur_ofs = (i64)readlane64((u64)post_ofs, f - 1);
					const i64 lm = (i64)readlane64((u64)post_lm, f - 1);
					if (applied) {
						R.last_match = lm;
						if (lm > p_skip)
							p_skip = lm;
					}
				}
				const bool used = applied > 0 || __ballot(committed && (L.pot || soft_me)) != 0;
				dense_used += used ? 1 : 0;
				if (++dense_rounds == DENSE_WINDOW) {
					if (dense_used < DENSE_WINDOW / 8 && !(batch_mode & 32))
						leave = 5; // seven rounds in eight had no use for this variant: four wavefronts do them faster
					dense_rounds = dense_used = 0;
				}
				if (!prof) {
					bump(8, applied);
					bump(9, __popcll(__ballot(committed && soft_me)));
					bump(10, 1);
				}
			}
			if (f < wcount && why_f >= 3 && why_f <= 7 && !(DENSE && prof && why_f == 4))
				bump(why_f, 1);
			inserts += n_ins;
			const i64 hc = R.hash_count + n_x;
			R.hash_count = hc < R.hash_limit ? hc : R.hash_limit;
			R.victim_round = (R.victim_round + n_ev) % (i64)R.max_chain;
			if (last_clean != 0xFFFFFFFFu) {
				R.clean_ptr = (i64)last_clean;
				R.tag_mask = better; // clean_one_from_hash() returns better_than_min
			}
			// (n_commit, not f: f also counts the candidates a match has jumped over)
			const bool poor = (batch_mode & 8) ? true : (batch_mode & 4) ? false : f < wcount && n_commit < POOR_COMMIT;
			if (poor) {
				if (++poor_rounds >= POOR_SCORE) {
					poor_rounds = POOR_SCORE - 2; // (two more poor rounds after the stretch and the next one follows)
					if (!DENSE && (batch_mode & 16) && serial_span > SPAN_MIN)
						leave = 4; // rounds were still poor after a first stretch of exact steps: the dense variant takes
							   // over from the next candidate on (host: scan_chunk_device)
					else {
						serial_left = serial_span;
						if (serial_span < SPAN_MAX)
							serial_span *= 2;
					}
				}
			} else {
				poor_rounds = poor_rounds > 2 ? poor_rounds - 2 : 0;
				if (!poor_rounds)
					serial_span = SPAN_MIN;
			}
		}
		if constexpr (DENSE) {
			// a kept lane with possible matches has lost its hits (the window's staging area overlays them), one whose
			// chain a committed eviction has changed reads other offsets now: both are simulated again
			// ... and so is one that was dead for this round's analysis only (inside a match the in-order pass emitted at
			// a lane the round did not get to): nobody checked its reads against the round's writes
			if (gi >= f && (L.pot || soft_me || !live))
				w_simd = false;
		}
		if (first_clean)
			w_simd = false; // the insert mask changed: every kept simulation is stale
		// simulations that read something a committed lane has just written are stale
		if (first_conf < f)
			w_simd = false;
		if (f < wcount) {
			if (why_f == 5) {
				// conflict only: lane f re-simulates against the updated table next round
				if (gi == f)
					w_simd = false;
				shift_window(f);
			} else {
				// complex / real match / sweep wrap / swept range: exact serial step (progress)
				if (master)
					serial_step(P_f, T_f);
				w_simd = false;
				shift_window(f + 1);
			}
		} else {
			shift_window(f);
		}
		lap(13); // bookkeeping, serial step, window shift
	}

#pragma unroll
	for (int d = 32; d >= 1; d >>= 1)
		miss_acc += (i64)bcast64((u64)miss_acc, lane ^ d);
	if (lane == 0)
		atomicAdd((unsigned long long *)&x_miss, (unsigned long long)miss_acc);
	__syncthreads();
	if (threadIdx.x == 0) {
		R.tag_misses += x_miss;
		st->p_skip = p_skip;
		st->last_match = R.last_match;
		st->cur_p = cur_p;
		st->cur_ofs = cur_ofs;
		st->cur_len = cur_len;
		st->tag_mask = R.tag_mask;
		st->min_mask = R.min_mask;
		st->hash_count = R.hash_count;
		st->clean_ptr = R.clean_ptr;
		st->victim_round = R.victim_round;
		st->n_records = n_rec;
		st->error = error;
		st->ext_p = R.ext_p;
		st->ext_op = R.ext_op;
		st->ext_done = R.ext_done;
		st->inserts = inserts;
		st->lookups = lookups;
		st->tag_hits = R.tag_hits;
		st->tag_misses = R.tag_misses;
		for (int k = 0; k < 16; k++)
			st->dbg[k] += dbg[k] + (k == 2 ? serial_n : 0);
	}
}
