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
