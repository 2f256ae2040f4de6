// bt_tiers.h -- which launch walks a hash bucket of the BT4 finder, by its number of positions (host side of lzma_mf.hip;
// no HIP types, so that a CPU test can include it).
//
// The buckets are sorted by length, longest first, and cut into consecutive launches:
//   launch 0            k_bt_wave<0>    a wavefront per bucket, tree nodes in memory     length >= min_len[0]
//   launch k = 1 .. n-1 k_bt_wave<CAP>  a wavefront per bucket, tree nodes in LDS        min_len[k] <= length < min_len[k-1],
//                                                                                       CAP = lds_cap[k] >= every such length
//   the rest            k_bt            a lane per bucket                                length < min_len[n-1]
#pragma once
#include <cstdint>

namespace lrzgpu {

constexpr int kBtTiers = 6;
struct BtTiers {
	uint32_t min_len[kBtTiers];
};
// node capacities k_bt_wave is instantiated for (32 B per node beside 34 KB of records and staging: 157 952 B of LDS at most)
constexpr uint32_t kBtLdsCap[kBtTiers - 1] = {3840, 2048, 1024, 512, 256};

struct BtLaunchPlan {
	BtTiers tiers;             // min_len[k] for k < n, 0xFFFFFFFF beyond
	uint32_t lds_cap[kBtTiers]; // 0 for launch 0
	int n;                     // launches of k_bt_wave (>= 1)
};

// long_min: buckets of at least this many positions keep their tree in memory (LRZGPU_BT_WAVE_MIN, default 4096).
// lds_min: shorter ones down to this many keep it in LDS (LRZGPU_BT_LDS_MIN); 0 = no LDS launches.
inline BtLaunchPlan bt_plan_launches(uint32_t long_min, uint32_t lds_min)
{
	BtLaunchPlan p;
	if (long_min < 1)
		long_min = 1;
	if (lds_min && lds_min <= kBtLdsCap[0] && long_min > kBtLdsCap[0] + 1)
		long_min = kBtLdsCap[0] + 1; // what no LDS launch can hold goes to the kernel that works from memory
	for (int k = 0; k < kBtTiers; k++)
		p.lds_cap[k] = 0;
	p.n = 1;
	p.tiers.min_len[0] = long_min;
	if (lds_min && lds_min < long_min)
		for (int c = 0; c < kBtTiers - 1; c++) {
			const uint32_t top = p.tiers.min_len[p.n - 1] - 1; // longest bucket still without a launch
			if (top > kBtLdsCap[c] || top < lds_min)
				continue; // capacity too small for it / nothing left to take
			if (c + 1 < kBtTiers - 1 && top <= kBtLdsCap[c + 1])
				continue; // the next smaller capacity holds them all
			const uint32_t below = c + 1 < kBtTiers - 1 ? kBtLdsCap[c + 1] + 1 : 1;
			p.lds_cap[p.n] = kBtLdsCap[c];
			p.tiers.min_len[p.n] = below > lds_min ? below : lds_min;
			p.n++;
			if (p.tiers.min_len[p.n - 1] == lds_min)
				break;
		}
	for (int k = p.n; k < kBtTiers; k++)
		p.tiers.min_len[k] = 0xFFFFFFFFu;
	return p;
}

} // namespace lrzgpu
