"""Developer tool: the headline workload handed over as a HOST buffer (PCIe-inclusive rate for DESIGN.md)."""
import importlib.util, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import conftest
import torch
B = conftest.load_bindings()
spec = importlib.util.spec_from_file_location("lrz_bench", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "bench.py"))
bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
import ctypes as C
mib = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
n = mib << 20
buf = bench.make_cfg3(n, max(1 << 20, n // 16), 1, torch.device("cuda:0"), "alnum")
host = buf[:n].cpu().numpy()
del buf
torch.cuda.empty_cache()
cores = os.cpu_count(); phys = os.sysconf("SC_PHYS_PAGES") * os.sysconf("SC_PAGE_SIZE")
for rep in range(2):  # the second pass has the pools warm, like bench.py after its warm-up
    c = B.make_control(level=7, threads=cores, processors=cores, ramsize=phys, window=21, host_threads=int(bench.usable_cpus() + 0.5), gpu_slots=8)
    out = C.POINTER(C.c_ubyte)(); olen = C.c_int64()
    f = B.lib().lrzgpu_compress_buffer
    f.argtypes = [C.POINTER(B.Control), C.c_void_p, C.c_int64, C.POINTER(C.POINTER(C.c_ubyte)), C.POINTER(C.c_int64)]
    t = time.time()
    rc = f(C.byref(c), C.c_void_p(host.ctypes.data), n, C.byref(out), C.byref(olen))
    dt = time.time() - t
    C.CDLL(None).free(out)
    print("host (pageable) input, cfg3 %d MiB, -w 21, pass %d: rc %d  %.2f s  %.1f MB/s  out %d" % (mib, rep, rc, dt, mib / dt, olen.value), flush=True)
