"""Developer tool: the bench workload shape at other -L levels (round trip checked with the library decoder)."""
import hashlib, importlib.util, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import conftest
import torch
B = conftest.load_bindings()
spec = importlib.util.spec_from_file_location("lrz_bench", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "bench.py"))
bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
mib = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
n = mib << 20
buf = bench.make_cfg2(n, 5, torch.device("cuda:0"), "alnum")
want = hashlib.md5(buf[:n].cpu().numpy()).digest()
cores = os.cpu_count(); phys = os.sysconf("SC_PHYS_PAGES") * os.sysconf("SC_PAGE_SIZE")
for level in (1, 4, 5, 9):
    ctl = B.make_control(level=level, threads=cores, processors=cores, ramsize=phys, host_threads=int(bench.usable_cpus() + 0.5), gpu_slots=8)
    t = time.time()
    img, ctl = B.compress_device(buf.data_ptr(), n, ctl=ctl, copy=False)
    dt = time.time() - t
    back = B.decompress_buffer(img, host_threads=16)
    print("-L%d: %d MiB in %.2f s (%.1f MB/s) -> %d bytes, dict %d, bufsize %d, round trip %s" %
          (level, mib, dt, mib / dt, len(img), ctl.dictSize_used, ctl.stream_bufsize, hashlib.md5(back).digest() == want), flush=True)
    img.free()
