#!/usr/bin/env python3
"""Developer tool: ONE LZMA block above 1.2 GiB through the whole-file driver (-p1: block = file, src/stream.c:1316-1323),
byte for byte against the CPU path (oracle rzip/container + the reference's LzmaCompress on the one block: minutes).
usage: big_block_case.py [MiB, default 1280]"""
import hashlib, importlib.util, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import conftest
import torch
import oracle_lib as O
B = conftest.load_bindings()
spec = importlib.util.spec_from_file_location("lrz_bench", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
def _ceiling():
    import ctypes
    f = B.lib().lrzgpu_max_block_bytes
    f.restype = ctypes.c_int64
    return f(0)
mib = int(sys.argv[1]) if len(sys.argv) > 1 else 1280
n = mib << 20
buf = bench.text_like_torch(n, 7, torch.device("cuda:0"))
ram = 64 << 30
kw = dict(level=7, threads=1, processors=1, ramsize=ram)
plan, chunk = B.plan(n, **kw)
print("plan: stream_bufsize %d, chunk %d, ceiling of this device %d" % (plan.stream_bufsize, chunk, _ceiling()), flush=True)
os.environ["LRZGPU_TRACE"] = "1"
t = time.time()
img, ctl = B.compress_device(buf.data_ptr(), n, copy=False, host_threads=int(bench.usable_cpus() + 0.5), **kw)
dt = time.time() - t
os.environ["LRZGPU_TRACE"] = ""
got = hashlib.sha256(img.view()).hexdigest()
print("GPU path: %.1f s, image %d bytes, sha256 %s" % (dt, len(img), got), flush=True)
if "--no-cpu" not in sys.argv:
    data = buf[:n].cpu().numpy()
    t = time.time()
    want, fs = O.compress_buffer(data, compression_level=7, threads=1, processors=1, ramsize=ram, workers=2)
    print("CPU path: %.1f s, image %d bytes, sha256 %s, %d blocks of up to %d bytes" % (time.time() - t, len(want), hashlib.sha256(want).hexdigest(), fs.n_blocks, fs.stream_bufsize), flush=True)
    print("IDENTICAL" if hashlib.sha256(want).hexdigest() == got else "DIFFERENT")
