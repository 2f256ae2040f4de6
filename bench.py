#!/usr/bin/env python3
"""bench.py -- compress MB/s (input) at -L7 lzma on MI355X, with roofline and CPU baseline.

A "step" = one complete pass of the hot path (rzip scan + lz4 gate + LZMA match finder on the GPU,
LZMA parser/range coder on host threads, container assembly, whole-input MD5) over ONE synthetic file
that is already resident in HBM when the timed region starts; it produces the complete .lrz image.

Default workload = the configuration BASELINE.json's metric is quoted on (configs[2], SURVEY 8d cfg 3):
16 GiB built from a 1 GiB seeded word-list-text base block repeated 16 times, every copy after the
first with one seeded byte mutation per 64 KiB; `-L7 -w 21` => 7 rzip chunks of 2 202 009 600 B + one of
1 765 801 984 B, every chunk holding internal long-range redundancy.  It fits one 288 GB GPU, so this
is also the N=1 workload: the chunks are scanned concurrently on the one device.
`--workload cfg2` is BASELINE configs[1] (4 GiB, half text + identical copy, one chunk).

N>1 (one process per GPU, torch.distributed over RCCL): STRONG scaling of the same one file -- rank r
compresses chunks r, r+N, ... (lrzgpu_compress_chunks_dev), the finished chunk images are handed to
rank 0 (send/recv over xGMI: the chunk hand-off), rank 0 checks the victim_round chain, computes the
MD5 and lays out the one .lrz.  No collective on the data path besides that hand-off.

Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import glob
import hashlib
import importlib.util
import json
import os
import sys
import resource
import time

# eight hardware queues for the streams of the chunk scanners, the gate and the finder workers (32 oversubscribes the
# GPU's resident queues and halves every kernel, see bindings.py): must be set before HIP initialises
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def load_module(name, rel):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "lrzip-next_amd", rel))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def load_bindings():
    return load_module("lrzip_next_amd_bindings", "bindings.py")


class Profile(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("tag_scan_ms", "resolve_ms", "crc_ms", "gather_ms", "lz4_ms", "mf_bt_ms",
                                           "mf_total_ms")] + \
               [(n, C.c_int64) for n in ("tag_scan_launches", "resolve_launches", "crc_launches", "gather_launches",
                                          "lz4_launches", "mf_launches", "tag_scan_positions", "resolve_lookups",
                                          "resolve_inserts", "resolve_match_bytes", "crc_bytes", "gather_bytes",
                                          "lz4_bytes", "mf_positions", "mf_entries")] + \
               [("scan_wall_ms", C.c_double), ("resolve_dbg", C.c_int64 * 16), ("long_compare_ms", C.c_double),
                ("long_compare_launches", C.c_int64), ("long_compare_bytes", C.c_int64), ("spec_rollbacks", C.c_int64),
                ("spec_cancelled_blocks", C.c_int64), ("victim_rescans", C.c_int64),
                ("union_ms", C.c_double * 8), ("peak_concurrency", C.c_double * 8),
                ("pipeline_s", C.c_double * 8), ("mf_wave_dbg", C.c_int64 * 4), ("early_s", C.c_double * 4),
                ("shard_s", C.c_double * 6)]


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


def make_cfg2(n_bytes, seed, device, alphabet="alnum"):
    """50 % long-range redundant: first half seeded text, second half an identical copy."""
    import torch
    half = n_bytes // 2
    base = text_like_torch(half, seed, device, alphabet=alphabet)
    buf = torch.empty(n_bytes + 256, dtype=torch.uint8, device=device)  # 256 B of readable padding
    buf[:half] = base
    buf[half:2 * half] = base
    buf[2 * half:] = 0
    return buf


def make_cfg3(n_bytes, base_bytes, seed, device, alphabet="alnum", mutate_every=65536):
    """SURVEY 8d cfg 3: a seeded text base block repeated to n_bytes; every copy after the first gets
    one seeded byte mutation per `mutate_every` bytes (position inside the 64 KiB cell and value seeded)."""
    import torch
    base = text_like_torch(base_bytes, seed, device, alphabet=alphabet)
    buf = torch.empty(n_bytes + 256, dtype=torch.uint8, device=device)
    g = torch.Generator(device=device)
    g.manual_seed(seed + 7919)
    at, copy = 0, 0
    while at < n_bytes:
        k = min(base_bytes, n_bytes - at)
        buf[at:at + k] = base[:k]
        if copy > 0:
            cells = k // mutate_every
            if cells:
                pos = torch.arange(cells, device=device, dtype=torch.int64) * mutate_every + \
                    torch.randint(0, mutate_every, (cells,), generator=g, device=device)
                val = torch.randint(0, 256, (cells,), generator=g, device=device, dtype=torch.int64).to(torch.uint8)
                buf[at + pos] = val
        at += k
        copy += 1
    buf[n_bytes:] = 0
    return buf


# the translation unit of csrc/ each profiled kernel is compiled from
KERNEL_UNIT = {"k_resolve": "rzip_scan.hip", "k_tag_scan": "rzip_scan.hip", "k_long_compare": "rzip_scan.hip",
               "k_gather_runs": "rzip_scan.hip", "k_crc32_tiles": "rzip_scan.hip", "k_bt": "lzma_mf.hip",
               "k_lz4_size": "lz4_gate.hip"}


def _elf_section(path, name):
    """Bytes of one section of an ELF64 file (None if it has none of that name)."""
    import struct
    with open(path, "rb") as f:
        d = f.read()
    if d[:4] != b"\x7fELF" or d[4] != 2:
        return None
    shoff, = struct.unpack_from("<Q", d, 0x28)
    shentsize, shnum, shstrndx = struct.unpack_from("<HHH", d, 0x3A)
    def sh(i):
        return struct.unpack_from("<IIQQQQIIQQ", d, shoff + i * shentsize)
    str_off, str_size = sh(shstrndx)[4], sh(shstrndx)[5]
    names = d[str_off:str_off + str_size]
    for i in range(shnum):
        h = sh(i)
        nm = names[h[0]:names.index(b"\0", h[0])].decode()
        if nm == name:
            return d[h[4]:h[4] + h[5]]
    return None


def build_id(unit=None):
    """Identifies the DEVICE code a PMC summary was measured for: the gfx950 code objects themselves -- the
    .hip_fatbin section of every csrc/*.hip.o (or, with `unit`, of that unit's object alone).  HBM bytes per launch of a
    kernel are a property of its code object and of the workload key stored next to it; a change to host code -- the
    parser, the driver, a header only host code reads, the host half of a .hip file -- leaves the id alone (round 3
    hashed the sources and every header they include: host-only edits to lzma_enc.h invalidated the round's last PMC
    pass).  Falls back to the sources when an object is missing (a tree that was never built)."""
    src = os.path.join(ROOT, "lrzip-next_amd", "csrc")
    units = [unit] if unit else sorted(os.path.basename(p) for p in glob.glob(os.path.join(src, "*.hip")))
    h = hashlib.sha256()
    for u in units:
        h.update(u.encode())
        obj = os.path.join(src, u + ".o")
        fat = _elf_section(obj, ".hip_fatbin") if os.path.exists(obj) else None
        if fat is None:
            h.update(b"source:")
            h.update(open(os.path.join(src, u), "rb").read())
        else:
            h.update(fat)
    return h.hexdigest()[:16]


def unit_build_ids():
    src = os.path.join(ROOT, "lrzip-next_amd", "csrc")
    return {os.path.basename(p): build_id(os.path.basename(p)) for p in sorted(glob.glob(os.path.join(src, "*.hip")))}


def pmc_traffic(kernel, workload_key):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC summary -- only if it was collected
    for exactly this build and workload (tools/pmc_collect.py writes the summary); else None."""
    path = os.path.join(ROOT, "profiles", "pmc_summary.json")
    try:
        d = json.load(open(path))
    except (OSError, ValueError):
        return None, "no PMC summary committed (tools/pmc_collect.py)"
    same_unit = ""
    if d.get("build_id") != build_id():
        # another build: the figure still stands if the code object this kernel comes from is byte-identical
        unit = KERNEL_UNIT.get(kernel)
        if not unit or d.get("unit_build_ids", {}).get(unit) != build_id(unit):
            return None, "PMC summary is for build %s, this is build %s: not reported" % (d.get("build_id"), build_id())
        same_unit = "; measured on build %s, this is build %s with the code object of %s unchanged" % (d.get("build_id"), build_id(), unit)
    if d.get("workload_key") != workload_key:
        return None, "PMC summary is for workload %s" % d.get("workload_key")
    ks = d.get("kernels", {})
    # (the resolver runs as k_resolve_mw<NW> on 2 / 4 wavefronts; the BT walk as k_bt + k_bt_wave per block)
    k = ks.get(kernel + "_mw") or ks.get(kernel + "+" + kernel + "_wave") or ks.get(kernel)
    if not k:
        return None, "kernel not in the PMC summary"
    return int(k["bytes_per_launch"]), d.get("note", "") + same_unit


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


def gpu_state():
    """Clocks / power / performance level of GPU 0 as rocm-smi reports them right now (context for run-to-run
    spread: the same launches took 378 ms on one box and 544 ms on another in round 2)."""
    import subprocess
    try:
        out = subprocess.run(["rocm-smi", "-d", "0", "--showclocks", "--showpower", "--showperflevel", "--showtemp", "--json"],
                             capture_output=True, text=True, timeout=20).stdout
        d = json.loads(out)
        card = d[sorted(d)[0]]
        keep = {}
        for k, v in card.items():
            kl = k.lower()
            if any(t in kl for t in ("sclk", "mclk", "fclk", "power", "performance level", "temperature (sensor junction)")):
                keep[k] = v
        return keep
    except Exception as e:  # no rocm-smi, no permission: say so
        return {"unavailable": repr(e)[:120]}


def file_to_file_leg(B, buf, n_bytes, fresh_ctl, passes=3):
    """The metric as test/speedtest.sh:98 defines it -- file in, file out: the identical bytes as a FILE (tmpfs, so
    the page cache is what the reference's mmap would read too), lrzgpu_compress_file() from its fd to the fd of an
    output file on the same tmpfs.  Input pages go host -> HBM over PCIe inside the timed region, the image is
    written with write().  One warm-up pass, then `passes` timed ones: the MEDIAN is the figure."""
    import torch
    best = None
    for d in ("/dev/shm", "/tmp"):
        try:
            st = os.statvfs(d)
            if st.f_bavail * st.f_frsize > n_bytes * 1.7:
                best = d
                break
        except OSError:
            pass
    if best is None:
        return {"skipped": "no tmpfs with room for input + image"}
    src, dst = os.path.join(best, "lrzgpu_bench_in.bin"), os.path.join(best, "lrzgpu_bench_out.lrz")
    try:
        with open(src, "wb") as f:
            piece = 1 << 30
            for o in range(0, n_bytes, piece):
                f.write(buf[o:min(n_bytes, o + piece)].cpu().numpy().tobytes())
        times = []
        size = 0
        for _ in range(1 + passes):
            fi = os.open(src, os.O_RDONLY)
            fo = os.open(dst, os.O_RDWR | os.O_CREAT | os.O_TRUNC, 0o644)
            c = fresh_ctl()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            rc = B.lib().lrzgpu_compress_file(C.byref(c), fi, fo)
            os.fsync(fo)
            dt = time.perf_counter() - t0
            os.close(fi)
            os.close(fo)
            if rc != 0:
                return {"error": rc}
            times.append(dt)
            size = os.path.getsize(dst)
        h = hashlib.sha256()
        with open(dst, "rb") as f:
            for piece in iter(lambda: f.read(64 << 20), b""):
                h.update(piece)
        digest = h.hexdigest()
        timed = sorted(times[1:])
        med = timed[len(timed) // 2] if len(timed) % 2 else 0.5 * (timed[len(timed) // 2 - 1] + timed[len(timed) // 2])
        return {"value": round(n_bytes / 1048576 / med, 2), "unit": "MB/s (2^20 B/s)", "seconds": round(med, 2),
                "timed_passes_seconds": [round(t, 2) for t in times[1:]], "first_pass_seconds": round(times[0], 2),
                "output_bytes": size, "image_sha256": digest,
                "what": "lrzgpu_compress_file(fd of a %d-byte file on %s -> fd of a file on %s): pread + H2D of every chunk, "
                        "write() of every chunk image, magic rewritten at the end; median of %d timed passes after one "
                        "warm-up pass" % (n_bytes, best, best, len(timed))}
    finally:
        for pth in (src, dst):
            try:
                os.unlink(pth)
            except OSError:
                pass


def cpu_baseline(buf, n_bytes, sample_bytes, ctl_kw, cores, desc):
    """Oracle driver (CPU restatement of rzip/lz4/container + the reference's own LZMA build, numThreads=2
    per block like the reference) on the HEAD of the identical buffer with the identical flags; chunking
    and block size are those of the whole file (oracle `file_size`), `cores` block workers."""
    import oracle_lib as O
    O.build()
    if O.ref_lzma() is None:
        return None
    data = buf[:sample_bytes].cpu().numpy()
    t0 = time.time()
    out, fs = O.compress_buffer(data, compression_level=ctl_kw["level"], threads=ctl_kw["threads"],
                                processors=ctl_kw["processors"], ramsize=ctl_kw["ramsize"], window=ctl_kw["window"],
                                workers=cores, file_size=n_bytes)
    dt = time.time() - t0
    return {"image_sha256": hashlib.sha256(out).hexdigest(), "image_bytes": len(out),
            "value": round(sample_bytes / 1048576 / dt, 2), "unit": "MB/s", "cores": cores, "kind": "port",
            "whole_file": bool(sample_bytes >= n_bytes),
            "sample": "the first %d bytes (%d whole rzip chunk(s)) of the IDENTICAL buffer (%s), identical flags and "
                      "block size (%d B, as for the whole file); oracle rzip/lz4/container restatement (one scan "
                      "thread, like the reference) + oracle/_ref LzmaCompress (reference LZMA sources, numThreads=2 "
                      "per block) on %d block workers; %.1f s; %d blocks"
                      % (sample_bytes, fs.n_chunks, desc, fs.stream_bufsize, cores, dt, fs.n_blocks),
            "seconds": round(dt, 2)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", choices=["cfg3", "cfg2"], default=os.environ.get("LRZGPU_BENCH_WORKLOAD", "cfg3"))
    ap.add_argument("--mib", type=int, default=int(os.environ.get("LRZGPU_BENCH_MIB", "0")),
                    help="file size in MiB (default: 16384 for cfg3, 4096 for cfg2)")
    ap.add_argument("--window", type=int, default=-1, help="-w (x100 MiB); default 21 for cfg3 (8 chunks of 16 GiB), unset for cfg2")
    ap.add_argument("--base-mib", type=int, default=0, help="cfg3 base block (default: file size / 16)")
    ap.add_argument("--cpu-sample-chunks", type=int, default=0,
                    help="rzip chunks of the same buffer the CPU baseline compresses (0 = the WHOLE file, the default)")
    ap.add_argument("--no-file-leg", action="store_true", help="skip the untimed-for-`value` file-to-file step")
    ap.add_argument("--file-passes", type=int, default=3, help="timed passes of the file-to-file leg (the median is reported)")
    ap.add_argument("--timeline", default="", help="write every kernel launch of the timed region as kernel,start_ms,end_ms (CSV) "
                                                   "plus a per-250-ms count of launches in flight per kernel")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--threads", type=int, default=0, help="-p (default: host cores)")
    ap.add_argument("--host-threads", type=int, default=0,
                    help="host LZMA encoder threads (default: the CPUs this process may use, cgroup quota included)")
    ap.add_argument("--alphabet", choices=sorted(ALPHABETS), default="alnum")
    ap.add_argument("--gpu-slots", type=int, default=8)
    ap.add_argument("--scan-slots", type=int, default=0)
    ap.add_argument("--verify", action="store_true", help="round-trip the last image through the library decoder (untimed)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # LRZGPU_BENCH_BACKEND=gloo: the same N-rank code path on a box with fewer GPUs than ranks (ranks share
    # devices, hand-off through host tensors) -- a functional check of the sharded step, not a measurement
    backend = os.environ.get("LRZGPU_BENCH_BACKEND", "nccl")
    local_dev = local_rank % max(1, torch.cuda.device_count()) if backend != "nccl" else local_rank
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_dev))
        else:
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
    torch.cuda.set_device(local_dev)
    dev = torch.device("cuda", local_dev)
    comm_dev = dev if backend == "nccl" else torch.device("cpu")

    B = load_bindings()
    SH = load_module("lrzip_next_amd_sharded", "sharded.py")
    L = B.lib()
    L.lrzgpu_profile_get.argtypes = [C.POINTER(Profile)]
    if L.lrzgpu_device_count() < 1:
        raise SystemExit("bench.py needs a GPU: liblrzgpu has no CPU fallback")

    cores = os.cpu_count() or 1
    threads = args.threads or cores  # -p as the reference defaults it on this host, the same at every N
    phys = os.sysconf("SC_PHYS_PAGES") * os.sysconf("SC_PAGE_SIZE")
    mib = args.mib or (16384 if args.workload == "cfg3" else 4096)
    n_bytes = mib << 20
    window = args.window if args.window >= 0 else (21 if args.workload == "cfg3" else 0)
    ctl_kw = dict(level=7, threads=threads, processors=cores, ramsize=phys, window=window)
    usable = usable_cpus()
    # the ranks of one node share the host: each gets its share of the usable CPUs for its encoders
    host_threads = args.host_threads or max(1, min(threads, int(usable + 0.5) // world))

    # every rank builds the same file (same seed): rank r needs its chunks resident, rank 0 the whole for the MD5
    if args.workload == "cfg3":
        base_bytes = (args.base_mib << 20) if args.base_mib else max(1 << 20, n_bytes // 16)
        buf = make_cfg3(n_bytes, base_bytes, 1, dev, args.alphabet)
        desc = ("%d MiB = a %d MiB seeded 5000-word text block (%d-symbol '%s' alphabet) repeated %d times, one seeded byte "
                "mutation per 64 KiB in every copy after the first" % (mib, base_bytes >> 20, len(ALPHABETS[args.alphabet]),
                                                                      args.alphabet, (n_bytes + base_bytes - 1) // base_bytes))
    else:
        buf = make_cfg2(n_bytes, 1, dev, args.alphabet)
        desc = ("%d MiB = seeded 5000-word text (%d-symbol '%s' alphabet) + identical copy at distance n/2"
                % (mib, len(ALPHABETS[args.alphabet]), args.alphabet))
    torch.cuda.synchronize()

    plan = B.make_control(**ctl_kw)
    chunk0 = C.c_int64()
    L.lrzgpu_plan(C.byref(plan), n_bytes, C.byref(chunk0))
    chunk_size = chunk0.value
    n_chunks = max(1, (n_bytes + chunk_size - 1) // chunk_size) if chunk_size else 1

    def fresh_ctl():
        return B.make_control(device=local_dev, host_threads=host_threads, gpu_slots=args.gpu_slots,
                              scan_slots=args.scan_slots, **ctl_kw)

    def step_single():
        out, ctl = B.compress_device(buf.data_ptr(), n_bytes, ctl=fresh_ctl(), copy=False)
        return out, ctl

    # ranks other than 0 keep only THEIR chunks resident (rank 0 hashes the whole input on its own thread from t = 0):
    # lrzgpu_compress_sharded_chunks_dev takes one device pointer per owned chunk
    chunk_ptrs = None
    if world > 1:
        # the hand-off's transport: the library's own, in C over RCCL (csrc/shard_rccl.cpp; torch.distributed only
        # carries the 128-byte unique id and the barriers of the measurement); gloo / LRZGPU_BENCH_TRANSPORT=torch:
        # the three callbacks over torch.distributed (lrzip-next_amd/sharded.py)
        transport = os.environ.get("LRZGPU_BENCH_TRANSPORT", "rccl-c" if backend == "nccl" else "torch")
        comm_close = None
        if transport == "rccl-c":
            def bcast_id(uid):
                t = torch.zeros(128, dtype=torch.uint8, device=comm_dev)
                if uid is not None:
                    t.copy_(torch.frombuffer(bytearray(uid), dtype=torch.uint8))
                dist.broadcast(t, src=0)
                return bytes(t.cpu().numpy().tobytes())
            try:
                comm, comm_close = SH.rccl_comm(L, rank, world, local_dev, bcast_id)
                ok = 1
            except Exception as e:  # (no librccl in the process, communicator refused ...): every rank must take the same way
                sys.stderr.write("bench.py rank %d: C transport not available (%s)\n" % (rank, e))
                ok = 0
            flag = torch.tensor([ok], dtype=torch.int32, device=comm_dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 0:
                if comm_close:
                    comm_close()
                    comm_close = None
                transport = "torch"
            comm_keep = None
        if transport != "rccl-c":
            comm, comm_keep = SH.torch_comm(rank, world, dist, torch, comm_dev)
        if rank != 0:
            mine = {}
            for k in range(rank, n_chunks, world):
                lo = k * chunk_size
                mine[k] = buf[lo:min(n_bytes, lo + chunk_size)].clone()
            buf = None  # (step_single is not used on this rank)
            torch.cuda.empty_cache()
            chunk_ptrs = [mine[k].data_ptr() if k in mine else 0 for k in range(n_chunks)]

    def step_sharded():
        """chunk k -> rank k % world, chain check, chunk images to rank 0 over RCCL, rank 0 lays out the file: all
        inside lrzgpu_compress_sharded*_dev (csrc/shard.cpp); torch.distributed is the transport it calls back into."""
        if chunk_ptrs is not None:
            return B.compress_sharded_dev(0, n_bytes, comm, ctl=fresh_ctl(), chunk_ptrs=chunk_ptrs)[:2]
        return B.compress_sharded_dev(buf.data_ptr(), n_bytes, comm, ctl=fresh_ctl())[:2]

    one_step = step_single if world == 1 else step_sharded

    for _ in range(args.warmup):
        one_step()

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    state_before = gpu_state() if rank == 0 else None
    L.lrzgpu_profile_reset()
    L.lrzgpu_profile_cpu.argtypes = [C.POINTER(C.c_double), C.c_int]
    L.lrzgpu_profile_cpu(None, 1)
    fence()
    ru0 = resource.getrusage(resource.RUSAGE_SELF)
    t0 = time.perf_counter()
    out = ctl = None
    # per step (rank 0): wall seconds and the accumulated device milliseconds of the two heaviest kernels -- a drift over
    # the steps of a long run (clocks, temperature) shows here, not only in the average
    per_step = []
    snap = Profile()
    t_prev, r_prev, b_prev, rl_prev, e_prev = t0, 0.0, 0.0, 0, 0.0
    cpu_snap = (C.c_double * 8)()
    for _ in range(args.steps):
        out, ctl = one_step()
        if rank == 0:
            L.lrzgpu_profile_get(C.byref(snap))
            L.lrzgpu_profile_cpu(cpu_snap, 0)
            t_now = time.perf_counter()
            per_step.append({"s": round(t_now - t_prev, 3), "k_resolve_ms": round(snap.resolve_ms - r_prev, 1),
                             "k_resolve_avg_launch_ms": round((snap.resolve_ms - r_prev) / max(1, snap.resolve_launches - rl_prev), 1),
                             "k_bt_ms": round(snap.mf_bt_ms - b_prev, 1), "encoders_cpu_s": round(cpu_snap[0] - e_prev, 1)})
            t_prev, r_prev, b_prev, rl_prev, e_prev = t_now, snap.resolve_ms, snap.mf_bt_ms, snap.resolve_launches, cpu_snap[0]
    fence()
    dt = time.perf_counter() - t0
    ru1 = resource.getrusage(resource.RUSAGE_SELF)
    cpu_s = (ru1.ru_utime - ru0.ru_utime) + (ru1.ru_stime - ru0.ru_stime)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=comm_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    prof = Profile()
    L.lrzgpu_profile_get(C.byref(prof))
    role_cpu = (C.c_double * 8)()
    L.lrzgpu_profile_cpu(role_cpu, 0)
    rank_stages = None
    if world > 1:
        # what each stage needs on a rank's own resources, per step: the MAX over the ranks (and rank 0's hash, which no
        # other rank has) -- the step cannot be shorter than the largest of them
        k = max(args.steps, 1)
        mine_s = [prof.pipeline_s[0] / max(host_threads, 1) / k, prof.pipeline_s[4] / k,
                  (prof.pipeline_s[2] + prof.pipeline_s[3]) / max(args.gpu_slots, 1) / k,
                  prof.shard_s[0] / k, prof.shard_s[1] / k, prof.shard_s[2] / k, prof.shard_s[3] / k, prof.shard_s[4] / k,
                  prof.shard_s[5] / k, prof.pipeline_s[1] / k, role_cpu[0] / k]
        t = torch.tensor(mine_s, dtype=torch.float64, device=comm_dev)
        tsum = t.clone()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        rank_stages = ([float(v) for v in t.tolist()], [float(v) for v in tsum.tolist()])
    state_after = gpu_state() if rank == 0 else None
    if rank == 0 and args.timeline:
        names = ["k_tag_scan", "k_resolve", "k_crc32_tiles", "k_gather_runs", "k_lz4_size", "k_bt", "finder", "k_long_compare"]
        L.lrzgpu_profile_intervals.argtypes = [C.c_int, C.POINTER(C.c_double), C.c_int]
        rows = []
        for kind, nm in enumerate(names):
            cnt = L.lrzgpu_profile_intervals(kind, None, 0)
            if cnt > 0:
                arr = (C.c_double * (2 * cnt))()
                L.lrzgpu_profile_intervals(kind, arr, cnt)
                rows += [(nm, arr[2 * i], arr[2 * i + 1]) for i in range(cnt)]
        rows.sort(key=lambda r: r[1])
        with open(args.timeline, "w") as f:
            f.write("kernel,start_ms,end_ms\n")
            for nm, a, b in rows:
                f.write("%s,%.3f,%.3f\n" % (nm, a, b))
            end = max([r[2] for r in rows] + [0.0])
            f.write("\n# launches in flight at t (ms since the profile reset), sampled every 250 ms\n# t_ms," + ",".join(names) + "\n")
            t = 0.0
            while t <= end:
                f.write("# %.0f," % t + ",".join(str(sum(1 for r in rows if r[0] == nm and r[1] <= t < r[2])) for nm in names) + "\n")
                t += 250.0

    if rank == 0:
        total_mib = args.steps * (n_bytes / 1048576)  # one file per step, whatever the number of GPUs
        value = total_mib / dt
        steps = max(args.steps, 1)
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
        # The dominant kernel = most accumulated device time (HIP events on the launching streams).  Several
        # launches of it run side by side (one resolver per chunk, one finder per GPU slot), so the sum can
        # exceed the wall time; `achieved` is per launch, as the contract asks.
        KIND = {"k_tag_scan": 0, "k_resolve": 1, "k_crc32_tiles": 2, "k_gather_runs": 3, "k_lz4_size": 4, "k_bt": 5,
                "finder (keys + sorts + k_bt + k_gather)": 6, "k_long_compare": 7}
        dom = max(kernels, key=lambda k: kernels[k][0])
        ms, launches, alg_bytes = kernels[dom]
        avg_ms = ms / max(launches, 1)
        achieved = (alg_bytes / max(launches, 1)) / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        workload_key = "%s-%dMiB-w%d-%s" % (args.workload, mib, window, args.alphabet)
        traffic, traffic_note = pmc_traffic(dom, workload_key)
        # whole path against the HBM roofline, SURVEY 8(d): 2N + 16C + 2M + 3L + 13 L_lzma per file
        n_tot = steps * n_bytes if world == 1 else None
        whole = None
        if world == 1:
            lit = prof.gather_bytes
            alg_path = 2 * n_tot + 16 * (prof.resolve_lookups + prof.resolve_inserts) + 2 * prof.resolve_match_bytes + 3 * lit + 13 * prof.mf_positions
            whole = {"bytes_alg_per_step": int(alg_path / steps), "frac_whole_path": round(alg_path / dt / 8e12, 6),
                     "formula": "2N + 16(lookups+inserts) + 2M + 3L + 13 L_lzma over the step time against 8 TB/s"}
        roofline = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 3), "peak": 8000.0, "unit": "GB/s",
                    "frac": round(achieved / 8000.0, 6), "traffic": traffic, "traffic_note": traffic_note,
                    "avg_launch_ms": round(avg_ms, 3), "launches": int(launches),
                    "launches_per_step": round(launches / steps, 1),
                    "algorithmic_bytes_per_launch": int(alg_bytes / max(launches, 1)),
                    "per_kernel_ms_per_step": {k: round(v[0] / steps, 2) for k, v in kernels.items()},
                    "per_step": per_step,
                    "per_kernel_GBps": {k: (round(v[2] / (v[0] * 1e-3) / 1e9, 3) if v[0] > 0 else 0.0)
                                        for k, v in kernels.items()},
                    # launches of one kernel run side by side (a resolver per chunk, a finder per GPU slot): the sums
                    # above exceed the step.  Wall time they COVER per step (union of their [start, end) intervals, HIP
                    # events against one base event) and the most that overlapped:
                    "per_kernel_wall_union_ms_per_step": {k: round(prof.union_ms[i] / steps, 2) for k, i in KIND.items()},
                    "per_kernel_peak_concurrency": {k: int(prof.peak_concurrency[i]) for k, i in KIND.items()},
                    "scan_wall_ms_per_step_summed_over_chunks": round(prof.scan_wall_ms / steps, 1),
                    "whole_path": whole,
                    "build_id": build_id(),
                    "resolver": dict(zip(("batches", "committed", "serial_steps", "stop_complex", "stop_match",
                                          "stop_conflict", "stop_novictim", "stop_sweptrange"),
                                         [int(v) for v in prof.resolve_dbg[:8]])),
                    "victim_rescans": int(prof.victim_rescans), "spec_rollbacks": int(prof.spec_rollbacks)}
        if world > 1:
            roofline["note"] = "kernel figures are rank 0's share of the chunks"
        # what bounds the step: the host stage (parser + range coder on `host_threads` threads) or a device stage
        pl = [v / steps for v in prof.pipeline_s]
        step_s = dt / steps
        stages = {"host parser + range coder (%d threads)" % host_threads: pl[0] / max(host_threads, 1),
                  "rzip scan (last chunk done)": pl[4],
                  "match finder + list copy (%d GPU slots)" % args.gpu_slots: (pl[2] + pl[3]) / max(args.gpu_slots, 1)}
        crit = max(stages, key=lambda k: stages[k])
        critical_path = {"stage": crit, "seconds_per_step": {k: round(v, 2) for k, v in stages.items()},
                         "step_seconds": round(step_s, 2),
                         "encoders_busy_s_per_step": round(pl[0], 1), "encoders_idle_s_per_step": round(pl[1], 1),
                         "finder_workers_busy_s_per_step": round(pl[2], 1), "lists_d2h_s_per_step": round(pl[3], 1),
                         "last_finder_done_s": round(pl[5], 2), "last_encode_done_s": round(pl[6], 2),
                         "first_encoder_start_s": round(prof.early_s[0] / steps, 2),
                         "early_start": {"blocks_started_before_complete_per_step": round(prof.early_s[2] / steps, 1),
                                         "finder_runs_on_their_prefixes_per_step": round(prof.early_s[3] / steps, 1),
                                         "encoder_s_waiting_inside_a_started_block_per_step": round(prof.early_s[1] / steps, 1)},
                         "note": "a stage's figure is the time it needs on its own resources (busy seconds / workers; the scans "
                                 "run unthrottled from t = 0); the step cannot be shorter than the largest.  The finder "
                                 "workers are throttled by the encoders (a bounded number of blocks hold lists in host "
                                 "memory), so their LAST block ends late whatever their speed"}
        if world > 1:
            mx, sm = rank_stages
            st_n = {"host parser + range coder (%d threads per rank, max over ranks)" % host_threads: mx[0],
                    "rzip scan (a rank's last chunk done, max over ranks)": mx[1],
                    "match finder + list copy (%d GPU slots per rank, max over ranks)" % args.gpu_slots: mx[2],
                    "whole-input hash on rank 0 (done at, since the start of its run)": mx[7],
                    "chunk hand-off into rank 0 (receive + lay-out)": mx[6]}
            critical_path = {"stage": max(st_n, key=lambda q: st_n[q]), "seconds_per_step": {q: round(v, 2) for q, v in st_n.items()},
                             "step_seconds": round(step_s, 2),
                             "a_rank_own_chunks_wall_s_max": round(mx[3], 2), "chain_check_wait_s_max": round(mx[4], 2),
                             "chunks_redone_s_max": round(mx[5], 2), "protocol_wall_s_max": round(mx[8], 2),
                             "encoders_cpu_s_per_step_all_ranks": round(sm[10], 1), "encoders_idle_s_per_step_all_ranks": round(sm[9], 1),
                             "host_cpus_usable_shared_by_the_ranks": round(usable, 1),
                             "encoder_cpu_s_over_usable_cpus": round(sm[10] / max(usable, 1.0), 2),
                             "note": "one file over N ranks is STRONG scaling of a format with two serial parts: the whole-input "
                                     "MD5 (one dependency chain, on rank 0 from t = 0: src/rzip.c:1195-1219) and the LZMA parse on "
                                     "host threads, which the N ranks of one node SHARE (encoder CPU-seconds / usable CPUs is a "
                                     "floor of the step at any N).  What shards is the GPU half: scan, gate, finder (DESIGN.md 7)"}
        file_leg = None
        if world == 1 and not args.no_file_leg:
            file_leg = file_to_file_leg(B, buf, n_bytes, fresh_ctl, max(1, args.file_passes))
        # the one kernel of the path that north_star expects at HBM speed, alone on the GPU (inside the step it runs on
        # a CU-masked scan stream beside everything else: per_kernel_GBps above): k_tag_scan over the first 2 GiB of the
        # same buffer at the mask the bench's scans mostly run under, through lrzgpu_tag_candidates_dev
        tag_scan_alone = None
        if world == 1:
            try:
                n_k1 = min(n_bytes, 2 << 30)
                cnt, _, ms1 = B.tag_candidates_dev(buf.data_ptr(), n_k1, min_mask=0x1ff, reps=40, only_tags=True, device=local_dev)  # (enough passes for the clocks of an idle GPU to come up)
                _, _, ms3 = B.tag_candidates_dev(buf.data_ptr(), n_k1, min_mask=0x1ff, reps=20, only_tags=False, device=local_dev)
                tag_scan_alone = {"kernel": "k_tag_scan", "bound": "hbm", "bytes": n_k1, "min_mask": "0x1ff", "candidates": int(cnt),
                                  "ms_per_pass": round(ms1, 3), "achieved": round(n_k1 / ms1 / 1e6, 1), "peak": 8000.0, "unit": "GB/s",
                                  "frac": round(n_k1 / ms1 / 1e6 / 8000.0, 4),
                                  "with_list_kernels_GBps": round(n_k1 / ms3 / 1e6, 1),
                                  "what": "1 B read per position (SURVEY 8d): positions per second of k_tag_scan alone on the idle GPU, "
                                          "average of the passes after the first; with k_tile_scan + k_compact_cands beside it"}
            except Exception as e:  # (never in the way of the line)
                tag_scan_alone = {"unavailable": repr(e)[:120]}
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            sample = n_bytes if args.cpu_sample_chunks <= 0 else min(n_bytes, args.cpu_sample_chunks * chunk_size)
            cpu = cpu_baseline(buf, n_bytes, sample, ctl_kw, max(1, int(usable + 0.5)), desc)
        verified = None
        if args.verify and out is not None:
            back = B.decompress_buffer(out, host_threads=max(1, int(usable + 0.5)))
            verified = bool(torch.equal(torch.frombuffer(bytearray(back), dtype=torch.uint8), buf[:n_bytes].cpu()))
        line = {
            "metric": "compress MB/s (input) at -L7 lzma", "value": round(value, 2), "unit": "MB/s (2^20 B/s)",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt * 1000 / steps, 1),
            "higher_is_better": True, "scaling": "strong" if world > 1 else "weak", "vs_baseline": None, "dtype": "u8",
            "data": "synthetic",
            "config": {"workload": "%s: %s; -L7 lzma%s => %d rzip chunk(s) of up to %d B; input resident in HBM; one "
                                   ".lrz image per step" % (args.workload, desc, " -w %d" % window if window else "", n_chunks, chunk_size),
                       "flags": "-L7 -p%d%s (PROCESSORS=%d, ramsize=%d)" % (threads, " -w %d" % window if window else "", cores, phys),
                       "stream_bufsize": int(ctl.stream_bufsize), "dict_size": int(ctl.dictSize_used),
                       "output_bytes": len(out), "host_threads": host_threads, "host_cpus_usable": round(usable, 1),
                       "host_cpu_seconds_rank0": round(cpu_s, 1),
                       "host_cpu_seconds_per_step_by_thread_role": dict(
                           [(nm, round(role_cpu[i] / steps, 1)) for i, nm in enumerate(("encoders (parser + range coder)", "gpu workers", "scanners", "whole-input hash", "reader"))] +
                           [("everything else (committer, Python, HIP runtime threads)", round((cpu_s - sum(role_cpu)) / steps, 1))]),
                       "parallelism": ("%d chunks scanned concurrently on 1 GPU" % n_chunks) if world == 1 else
                                      ("chunk k -> GPU k mod %d, chunk images handed to rank 0 over %s send/recv" % (world, ("RCCL (C transport, csrc/shard_rccl.cpp)" if transport == "rccl-c" else "RCCL via torch.distributed") if backend == "nccl" else backend))},
            "roofline": roofline, "roofline_k_tag_scan_alone": tag_scan_alone, "cpu_baseline": cpu,
            "critical_path": critical_path,
            "value_file_to_file": file_leg,
            "value_note": "`value` = the K timed steps with the input ALREADY RESIDENT in HBM -- what this run's measurement contract "
                          "prescribes for `value` (the PCIe-inclusive rate is never `value`); the metric exactly as "
                          "test/speedtest.sh:98 / SURVEY 8(d) define it (file open -> magic written) is value_file_to_file.value, "
                          "the median of its timed passes, and vs_cpu_baseline.file_to_file is the ratio to quote",
            "gpu_state": {"before_timed_region": state_before, "after_timed_region": state_after},
        }
        if cpu and cpu.get("value"):
            line["vs_cpu_baseline"] = {"hbm_resident": round(value / cpu["value"], 2),
                                       "file_to_file": round(file_leg["value"] / cpu["value"], 2) if file_leg and file_leg.get("value") else None,
                                       "cpu_sample_is_whole_file": cpu.get("whole_file")}
        if verified is not None:
            line["round_trip_ok"] = verified
        # byte-for-byte parity on the measured configuration itself: the image of the last timed step, the image the
        # file-to-file leg wrote and the image the CPU path (oracle + the reference's LZMA build) made of the identical
        # buffer with the identical flags -- compared by digest, at full size, in every run
        if out is not None:
            mine = hashlib.sha256(out.view() if hasattr(out, "view") else out).hexdigest()
            par = {"gpu_image_sha256": mine, "gpu_image_bytes": len(out)}
            if file_leg and file_leg.get("image_sha256"):
                par["file_to_file_image_sha256"] = file_leg["image_sha256"]
                par["file_to_file_identical_to_hbm_resident"] = file_leg["image_sha256"] == mine
            if cpu and cpu.get("image_sha256") and cpu.get("whole_file"):
                par["cpu_image_sha256"] = cpu["image_sha256"]
                line["identical_to_cpu_baseline"] = cpu["image_sha256"] == mine and cpu["image_bytes"] == len(out)
            else:
                line["identical_to_cpu_baseline"] = None  # (the CPU leg compressed a sample of the file, or did not run)
            line["parity"] = par
        print(json.dumps(line), flush=True)
    if world > 1:
        if comm_close:
            comm_close()
        dist.destroy_process_group()
    # parked buffers, workspaces and streams go back before the process ends (a profiler wrapped around this
    # command wants to see every queue closed)
    del out
    torch.cuda.synchronize()
    L.lrzgpu_shutdown.restype = None
    L.lrzgpu_shutdown()


if __name__ == "__main__":
    main()
