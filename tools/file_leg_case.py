#!/usr/bin/env python3
"""Developer tool: the headline workload file to file (lrzgpu_compress_file on /dev/shm) beside the same bytes resident in
HBM, with the driver's timeline (LRZGPU_TRACE=1 lines "lrzgpu reader" / "lrzgpu driver") -- where the file leg loses."""
import ctypes as C, importlib.util, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import conftest
import torch
B = conftest.load_bindings()
spec = importlib.util.spec_from_file_location("lrz_bench", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
mib = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
n = mib << 20
dev = torch.device("cuda:0")
buf = bench.make_cfg3(n, max(1 << 20, n // 16), 1, dev, "alnum")
cores = os.cpu_count(); phys = os.sysconf("SC_PHYS_PAGES") * os.sysconf("SC_PAGE_SIZE")
def ctl():
    return B.make_control(level=7, threads=cores, processors=cores, ramsize=phys, window=21, host_threads=int(bench.usable_cpus() + 0.5), gpu_slots=8)
src, dst = "/dev/shm/lrzgpu_case_in.bin", "/dev/shm/lrzgpu_case_out.lrz"
with open(src, "wb") as f:
    for o in range(0, n, 1 << 30):
        f.write(buf[o:min(n, o + (1 << 30))].cpu().numpy().tobytes())
try:
    for rep in range(3):
        os.environ["LRZGPU_TRACE"] = "1" if rep else ""
        t = time.perf_counter(); out, c = B.compress_device(buf.data_ptr(), n, ctl=ctl(), copy=False); dt = time.perf_counter() - t
        print("HBM resident pass %d: %.2f s  %.1f MB/s" % (rep, dt, mib / dt), flush=True)
        del out
        fi = os.open(src, os.O_RDONLY); fo = os.open(dst, os.O_RDWR | os.O_CREAT | os.O_TRUNC, 0o644)
        cc = ctl()
        t = time.perf_counter(); rc = B.lib().lrzgpu_compress_file(C.byref(cc), fi, fo); t1 = time.perf_counter(); os.fsync(fo); dt = time.perf_counter() - t
        os.close(fi); os.close(fo)
        print("file to file pass %d: rc %d  %.2f s (%.2f before fsync)  %.1f MB/s" % (rep, rc, dt, t1 - t, mib / dt), flush=True)
finally:
    for p in (src, dst):
        try: os.unlink(p)
        except OSError: pass
