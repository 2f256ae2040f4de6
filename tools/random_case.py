"""Developer tool: the incompressible case (BASELINE config 5 shape, smaller): random bytes resident in HBM."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import conftest
import torch
B = conftest.load_bindings()
mib = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
n = mib << 20
g = torch.Generator(device="cuda"); g.manual_seed(7)
buf = torch.randint(0, 256, (n + 4096,), dtype=torch.uint8, device="cuda", generator=g)
torch.cuda.synchronize()
cores = os.cpu_count()
phys = os.sysconf("SC_PHYS_PAGES") * os.sysconf("SC_PAGE_SIZE")
ctl = B.make_control(level=7, threads=cores, processors=cores, ramsize=phys, host_threads=16, gpu_slots=8)
t = time.time()
img, ctl = B.compress_device(buf.data_ptr(), n, ctl=ctl, copy=False)
dt = time.time() - t
print("random %d MiB: %.2f s  %.1f MB/s  out %d" % (mib, dt, mib / dt, len(img)))
import ctypes as C, importlib.util
spec = importlib.util.spec_from_file_location("bench", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "bench.py")); bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
p = bench.Profile(); L = B.lib(); L.lrzgpu_profile_get.argtypes = [C.POINTER(bench.Profile)]; L.lrzgpu_profile_get(C.byref(p))
print("resolver: %.1f s in %d launches; lookups %d inserts %d; %s" % (p.resolve_ms / 1e3, p.resolve_launches, p.resolve_lookups, p.resolve_inserts, dict(zip(
    ("batches", "committed", "serial_steps", "stop_complex", "stop_match", "stop_conflict", "stop_novictim", "stop_sweptrange"), [int(v) for v in p.resolve_dbg[:8]]))))
print("tag scan %.1f s, crc %.1f s, gather %.1f s, lz4 %.1f s (summed)" % (p.tag_scan_ms / 1e3, p.crc_ms / 1e3, p.gather_ms / 1e3, p.lz4_ms / 1e3))
