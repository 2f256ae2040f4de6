#!/usr/bin/env python3
"""bench.py -- compress MB/s (input) at -L7 lzma on MI355X, with roofline and CPU baseline.

A "step" = one complete pass of the hot path (rzip scan + lz4 gate + LZMA match finder on the GPU,
LZMA parser/range coder on host threads, container assembly) over one synthetic buffer that is
already resident in HBM when the timed region starts.

N=1 workload = BASELINE.json configs[1]: 4 GiB synthetic 50 %-long-range-redundant buffer
(2 GiB seeded word-list text followed by an identical copy), -L7 lzma, single rzip chunk.
N>1: one process per GPU (torch.distributed, backend nccl = RCCL), every rank compresses its own
buffer of that shape -- independent rzip chunks, no data-path collective (weak scaling); the only
collectives are the bracketing barriers and the MAX-reduce of the elapsed time.

Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import importlib.util
import json
import os
import sys
import resource
import time

# one hardware queue per stream (scan, gate, finder workers): must be set before HIP initialises
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def load_bindings():
    name = "lrzip_next_amd_bindings"
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "lrzip-next_amd", "bindings.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


class Profile(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("tag_scan_ms", "resolve_ms", "crc_ms", "gather_ms", "lz4_ms", "mf_bt_ms",
                                           "mf_total_ms")] + \
               [(n, C.c_int64) for n in ("tag_scan_launches", "resolve_launches", "crc_launches", "gather_launches",
                                          "lz4_launches", "mf_launches", "tag_scan_positions", "resolve_lookups",
                                          "resolve_inserts", "resolve_match_bytes", "crc_bytes", "gather_bytes",
                                          "lz4_bytes", "mf_positions", "mf_entries")] + \
               [("scan_wall_ms", C.c_double), ("resolve_dbg", C.c_int64 * 16), ("long_compare_ms", C.c_double),
                ("long_compare_launches", C.c_int64), ("long_compare_bytes", C.c_int64), ("spec_rollbacks", C.c_int64), ("spec_cancelled_blocks", C.c_int64)]


ALPHABETS = {
    # 62 symbols: what source code / logs / mixed text look like to rzip's XOR tag.  The tag of a
    # 31-byte window only depends on the PARITY of each byte value's count, so a 27-symbol alphabet
    # collapses the tag space to 2^27 values and every bucket saturates at max_chain_len entries
    # (the degenerate case of src/rzip.c's hash, kept available as --alphabet lower).
    "alnum": b"abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ0123456789",
    "lower": b"abcdefghijklmnopqrstuvwxyz",
}


def text_like_torch(n, seed, device, piece=256 << 20, alphabet="alnum"):
    """Seeded word-list pseudo text: 5000 words of 2..9 symbols, space separated."""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    nwords = 5000
    abc = torch.tensor(list(ALPHABETS[alphabet]), dtype=torch.uint8, device=device)
    wl = torch.randint(2, 10, (nwords,), generator=g, device=device)
    chars = abc[torch.randint(0, len(abc), (nwords, 10), generator=g, device=device)]
    out = torch.empty(n, dtype=torch.uint8, device=device)
    done = 0
    while done < n:
        want = min(piece, n - done)
        m = want // 6 + 4096  # mean word+space is 6.5 bytes
        idx = torch.randint(0, nwords, (m,), generator=g, device=device)
        lens = wl[idx] + 1
        off = torch.cumsum(lens, 0) - lens
        total = int(off[-1] + lens[-1])
        buf = torch.full((total,), 32, dtype=torch.uint8, device=device)
        for j in range(9):
            sel = (lens - 1) > j
            buf[off[sel] + j] = chars[idx[sel], j]
        take = min(want, total)
        out[done:done + take] = buf[:take]
        done += take
        del idx, lens, off, buf
    return out


def make_workload(n_bytes, seed, device, alphabet="alnum"):
    """50 % long-range redundant: first half seeded text, second half an identical copy."""
    import torch
    half = n_bytes // 2
    base = text_like_torch(half, seed, device, alphabet=alphabet)
    buf = torch.empty(n_bytes + 256, dtype=torch.uint8, device=device)  # 256 B of readable padding
    buf[:half] = base
    buf[half:2 * half] = base
    buf[2 * half:] = 0
    return buf


def cpu_baseline(sample_bytes, ctl_kw, cores, alphabet="alnum"):
    """Oracle driver (CPU restatement of rzip/lz4/container + the reference's own LZMA build) on a
    bounded sample of the same workload shape, all host cores as block-compression workers."""
    import oracle_lib as O
    import torch
    O.build()
    if O.ref_lzma() is None:
        return None
    data = bytes(make_workload(sample_bytes, 1, "cpu", alphabet)[:sample_bytes].numpy())
    t0 = time.time()
    out, fs = O.compress_buffer(data, compression_level=ctl_kw["level"], threads=ctl_kw["threads"],
                                processors=ctl_kw["processors"], ramsize=ctl_kw["ramsize"], workers=cores)
    dt = time.time() - t0
    return {"value": round(sample_bytes / 1048576 / dt, 2), "unit": "MB/s", "cores": cores, "kind": "port",
            "sample": "%d MiB of the same shape (half seeded text + identical copy), -L7, oracle rzip/lz4/container "
                      "restatement + oracle/_ref LzmaCompress (reference LZMA sources, numThreads=2) on %d worker "
                      "threads; %.1f s; %d blocks of %d B" % (sample_bytes >> 20, cores, dt, fs.n_blocks, fs.stream_bufsize),
            "seconds": round(dt, 2)}


def pmc_traffic(kernel, args, launches):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
    (profiles/r1_k_resolve_pmc.json; counters cannot be read from inside the process)."""
    path = os.path.join(ROOT, "profiles", "r1_k_resolve_pmc.json")
    try:
        d = json.load(open(path))
    except (OSError, ValueError):
        return None, "no PMC summary committed"
    if kernel != d.get("kernel") or args.mib != d.get("workload_mib") or args.alphabet != d.get("alphabet"):
        return None, "PMC summary is for %s on the %s MiB workload" % (d.get("kernel"), d.get("workload_mib"))
    # per launch of THIS run: the segment count (launches) may differ from the profiled build's
    b = (d["fetch_kb_total"] + d["write_kb_total"]) * 1024.0 / max(1, launches)
    return int(b), ("(FETCH_SIZE + WRITE_SIZE) x 1024 per launch from profiles/r1_bench4g_pmc_hbm.csv, raw; "
                    "gfx950 FETCH_SIZE under-reports wide streams 2x, so reads are between 1x and 2x the fetch part")


def usable_cpus():
    """CPUs this process can actually burn: affinity mask, capped by the cgroup CPU quota."""
    n = float(len(os.sched_getaffinity(0)))
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, float(txt[0]) / float(txt[1]))
            else:
                q = float(txt[0])
                if q > 0:
                    n = min(n, q / float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read()))
        except (OSError, ValueError, IndexError):
            pass
    return max(1.0, n)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1)
    ap.add_argument("--warmup", type=int, default=0)
    ap.add_argument("--mib", type=int, default=int(os.environ.get("LRZGPU_BENCH_MIB", "4096")),
                    help="workload size per GPU in MiB (default: the 4 GiB configuration)")
    ap.add_argument("--cpu-sample-mib", type=int, default=int(os.environ.get("LRZGPU_CPU_SAMPLE_MIB", "1024")))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--threads", type=int, default=0, help="-p (default: host cores)")
    ap.add_argument("--host-threads", type=int, default=0,
                    help="host LZMA encoder threads (default: the CPUs this process may use, cgroup quota included)")
    ap.add_argument("--alphabet", choices=sorted(ALPHABETS), default="alnum")
    ap.add_argument("--gpu-slots", type=int, default=8)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group(backend="nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    B = load_bindings()
    L = B.lib()
    L.lrzgpu_profile_get.argtypes = [C.POINTER(Profile)]
    if L.lrzgpu_device_count() < 1:
        raise SystemExit("bench.py needs a GPU: liblrzgpu has no CPU fallback")

    cores = os.cpu_count() or 1
    threads = args.threads or cores  # -p as the reference defaults it on this host, the same at every N
    phys = os.sysconf("SC_PHYS_PAGES") * os.sysconf("SC_PAGE_SIZE")
    ctl_kw = dict(level=7, threads=threads, processors=cores, ramsize=phys)
    usable = usable_cpus()
    # the ranks of one node share the host: each gets its share of the usable CPUs for its encoders
    host_threads = args.host_threads or max(1, min(threads, int(usable + 0.5) // world))
    n_bytes = args.mib << 20

    buf = make_workload(n_bytes, 1 + rank, dev, args.alphabet)
    torch.cuda.synchronize()

    def one_step():
        ctl = B.make_control(device=local_rank, host_threads=host_threads, gpu_slots=args.gpu_slots, **ctl_kw)
        out, ctl = B.compress_device(buf.data_ptr(), n_bytes, ctl=ctl, copy=False)
        return out, ctl

    for _ in range(args.warmup):
        one_step()

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    L.lrzgpu_profile_reset()
    fence()
    ru0 = resource.getrusage(resource.RUSAGE_SELF)
    t0 = time.perf_counter()
    out = ctl = None
    for _ in range(args.steps):
        out, ctl = one_step()
    fence()
    dt = time.perf_counter() - t0
    ru1 = resource.getrusage(resource.RUSAGE_SELF)
    cpu_s = (ru1.ru_utime - ru0.ru_utime) + (ru1.ru_stime - ru0.ru_stime)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    prof = Profile()
    L.lrzgpu_profile_get(C.byref(prof))

    if rank == 0:
        total_mib = args.steps * world * (n_bytes / 1048576)
        value = total_mib / dt
        # dominant kernel by accumulated device time
        kernels = {
            # k_resolve: one 16-byte slot per probe and per insert + both operands of the match bytes it
            # verifies itself (extents beyond 4 MiB are k_long_compare's)
            "k_resolve": (prof.resolve_ms, prof.resolve_launches,
                          16 * (prof.resolve_lookups + prof.resolve_inserts) + 2 * prof.resolve_match_bytes
                          - prof.long_compare_bytes),
            "k_long_compare": (prof.long_compare_ms, prof.long_compare_launches, prof.long_compare_bytes),
            "k_bt": (prof.mf_bt_ms, prof.mf_launches, 13 * prof.mf_positions),
            "k_tag_scan": (prof.tag_scan_ms, prof.tag_scan_launches, prof.tag_scan_positions),
            "k_lz4_size": (prof.lz4_ms, prof.lz4_launches, prof.lz4_bytes),
            "k_crc32_tiles": (prof.crc_ms, prof.crc_launches, prof.crc_bytes),
            "k_gather_runs": (prof.gather_ms, prof.gather_launches, 2 * prof.gather_bytes),
        }
        # The dominant kernel is the one on the step's critical path.  k_lz4_size / k_bt run a hundred
        # stream blocks concurrently on otherwise idle CUs, so their accumulated device time over-counts;
        # k_resolve is a single wavefront launched back to back on the scan stream for scan_wall_ms.
        dom = "k_resolve" if prof.resolve_ms > 0 else max(kernels, key=lambda k: kernels[k][0])
        ms, launches, alg_bytes = kernels[dom]
        avg_ms = ms / max(launches, 1)
        achieved = (alg_bytes / max(launches, 1)) / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        traffic, traffic_note = pmc_traffic(dom, args, launches)
        roofline = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 3), "peak": 8000.0, "unit": "GB/s",
                    "frac": round(achieved / 8000.0, 6), "traffic": traffic, "traffic_note": traffic_note,
                    "avg_launch_ms": round(avg_ms, 3), "launches": int(launches),
                    "algorithmic_bytes_per_launch": int(alg_bytes / max(launches, 1)),
                    "per_kernel_ms": {k: round(v[0], 2) for k, v in kernels.items()},
                    "per_kernel_GBps": {k: (round(v[2] / (v[0] * 1e-3) / 1e9, 3) if v[0] > 0 else 0.0)
                                        for k, v in kernels.items()},
                    "scan_wall_ms": round(prof.scan_wall_ms, 1),
                    "critical_path_frac": round(prof.scan_wall_ms / (dt * 1000.0), 3),
                    "resolver": dict(zip(("batches", "committed", "serial_steps", "stop_complex", "stop_match",
                                          "stop_conflict", "stop_novictim", "stop_sweptrange", "cyc_refill",
                                          "cyc_simulate", "cyc_victims", "cyc_conflict", "cyc_apply", "cyc_tail", "cyc_verify", "cyc_displace"),
                                         [int(v) for v in prof.resolve_dbg]))}
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            # workers: each oracle block worker runs the reference LzmaCompress with numThreads=2
            cpu = cpu_baseline(min(args.cpu_sample_mib << 20, n_bytes), ctl_kw, max(1, int(usable + 0.5)), args.alphabet)
        line = {
            "metric": "compress MB/s (input) at -L7 lzma", "value": round(value, 2), "unit": "MB/s (2^20 B/s)",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt * 1000 / args.steps, 1),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": "%d MiB synthetic 50%%-long-range-redundant buffer (seeded 5000-word text over the "
                                   "%d-symbol '%s' alphabet + identical copy at distance n/2), -L7 lzma, single rzip "
                                   "chunk per GPU, input resident in HBM" % (args.mib, len(ALPHABETS[args.alphabet]), args.alphabet),
                       "flags": "-L7 -p%d (PROCESSORS=%d, ramsize=%d)" % (threads, cores, phys),
                       "stream_bufsize": int(ctl.stream_bufsize), "dict_size": int(ctl.dictSize_used),
                       "output_bytes": len(out), "host_threads": host_threads, "host_cpus_usable": round(usable, 1),
                       "host_cpu_seconds": round(cpu_s, 1), "parallelism": "chunk-per-gpu x%d" % world},
            "roofline": roofline, "cpu_baseline": cpu,
        }
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
