"""Developer tool: the headline workload handed over as a HOST buffer (PCIe-inclusive rate for DESIGN.md)."""
import importlib.util, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import conftest
import torch
B = conftest.load_bindings()
spec = importlib.util.spec_from_file_location("lrz_bench", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "bench.py"))
bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
mib = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
n = mib << 20
buf = bench.make_workload(n, 1, torch.device("cuda:0"), "alnum")
host = buf[:n].cpu().numpy().tobytes()
del buf
torch.cuda.empty_cache()
cores = os.cpu_count(); phys = os.sysconf("SC_PHYS_PAGES") * os.sysconf("SC_PAGE_SIZE")
t = time.time()
img, ctl = B.compress_buffer(host, level=7, threads=cores, processors=cores, ramsize=phys, host_threads=int(bench.usable_cpus() + 0.5), gpu_slots=8)
dt = time.time() - t
print("host input %d MiB: %.2f s  %.1f MB/s  out %d" % (mib, dt, mib / dt, len(img)))
