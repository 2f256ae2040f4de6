"""Developer tool: one chunk beyond 2^32 bytes (5 GiB; 5-byte offsets, positions above 4 Gi) round trip."""
import hashlib, importlib.util, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import conftest
import torch
B = conftest.load_bindings()
spec = importlib.util.spec_from_file_location("lrz_bench", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "bench.py"))
bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
mib = int(sys.argv[1]) if len(sys.argv) > 1 else 5120
window = int(sys.argv[2]) if len(sys.argv) > 2 else 0  # -w (x100 MiB chunks), 0 = one chunk
n = mib << 20
buf = bench.make_cfg2(n, 3, torch.device("cuda:0"), "alnum")
want = hashlib.md5(buf[:n].cpu().numpy()).digest()
cores = os.cpu_count(); phys = os.sysconf("SC_PHYS_PAGES") * os.sysconf("SC_PAGE_SIZE")
ctl = B.make_control(level=7, threads=cores, processors=cores, ramsize=phys, window=window, host_threads=int(bench.usable_cpus() + 0.5), gpu_slots=8)
t = time.time()
img, ctl = B.compress_device(buf.data_ptr(), n, ctl=ctl, copy=False)
dt = time.time() - t
info_ok = bytes(ctl.hash_resblock) == want
back = B.decompress_buffer(img, host_threads=16)
print("%d MiB, -w %d: %.2f s (%.1f MB/s), image %d bytes, md5 in control %s, decode == input %s" %
      (mib, window, dt, mib / dt, len(img), info_ok, hashlib.md5(back).digest() == want and len(back) == n))
