#!/usr/bin/env python3
"""Build check: lzma_parser.cpp is built twice, for x86-64-v3 and for x86-64-v4, and both objects emit
the same weak symbols from shared headers; only the link order keeps AVX-512 code out of functions a non-AVX-512 host
executes.  This walks the disassembly of liblrzgpu.so and fails if any function outside namespace isa_v4 touches a
zmm or mask register."""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
so = os.path.join(ROOT, "lrzip-next_amd", "liblrzgpu.so")
out = subprocess.run(["objdump", "-d", "--no-show-raw-insn", "-C", so], capture_output=True, text=True).stdout
cur, bad = None, {}
for line in out.splitlines():
    m = re.match(r"^[0-9a-f]+ <(.*)>:$", line)
    if m:
        cur = m.group(1)
        continue
    if cur and (re.search(r"%zmm\d+", line) or re.search(r"%k[1-7]\b", line)):
        if "isa_v4" not in cur:
            bad[cur] = bad.get(cur, 0) + 1
if bad:
    print("AVX-512 instructions outside isa_v4:")
    for k, v in sorted(bad.items()):
        print("  %5d  %s" % (v, k))
    sys.exit(1)
print("ok: AVX-512 code only inside lrzgpu::isa_v4")
