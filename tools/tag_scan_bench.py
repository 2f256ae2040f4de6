"""k_tag_scan alone on the GPU: GB/s of positions (1 B read per position) by mask, against the 8 TB/s roofline.
usage: python tools/tag_scan_bench.py [MiB]   -- the bench text, one chunk in HBM"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
mib = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
B = bench.load_bindings()
n = mib << 20
buf = torch.empty(n + 256, dtype=torch.uint8, device="cuda")
buf[:n] = bench.text_like_torch(n, 12345, "cuda:0")
buf[n:] = 0
torch.cuda.synchronize()
for mask in (1, 0xF, 0x1FF, 0xFFF):
    for only in (True, False):
        cnt, chk, ms = B.tag_candidates_dev(buf.data_ptr(), n, min_mask=mask, reps=6, only_tags=only)
        print("%d MiB text, mask %#x: %d candidates (%.2f %%); %s: %.3f ms per pass = %.1f GB/s of positions (%.4f of 8 TB/s)"
              % (mib, mask, cnt, 100.0 * cnt / n, "k_tag_scan alone" if only else "k_tag_scan + k_tile_scan + k_compact_cands", ms, n / ms / 1e6, n / ms / 1e6 / 8000.0), flush=True)
