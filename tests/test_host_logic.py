"""CPU tests of the product's host logic and of the C ABI surface (no compute calls needing a GPU)."""
import ctypes as C
import hashlib
import os
import re
import sys

import pytest

import datagen

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol(B):
    import glob
    hdr = "".join(open(p).read() for p in sorted(glob.glob(os.path.join(ROOT, "include", "*.h"))))
    names = sorted(set(re.findall(r"\b(lrzgpu_[A-Za-z0-9_]+)\s*\(", hdr)))
    assert len(names) >= 23 and "lrzgpu_read_magic" in names
    L = B.lib()
    for n in names:
        assert hasattr(L, n), "missing export %s" % n


def test_compute_entry_points_fail_loudly_without_gpu(B):
    L = B.lib()
    if L.lrzgpu_device_count() > 0:
        pytest.skip("a GPU is visible")
    assert L.lrzgpu_lz4_compresses(b"x" * 100, 100, 100, 0) == -100  # LRZGPU_E_NODEVICE
    with pytest.raises(RuntimeError):
        B.compress_buffer(b"hello world" * 100)
    with pytest.raises(RuntimeError):
        B.hash_search(b"hello world" * 100)


@pytest.mark.parametrize("st_size", [0, 1, 5000, 10 << 20, 64 << 20, 700 << 20, 4 << 30, 16 << 30])
def test_plan_equals_oracle_sizing(B, O, st_size):
    """open_stream_out / rzip_fd sizing: product (stream_layer.cpp) vs oracle (container_oracle.c)."""
    import ctypes
    for level in (5, 7, 9):
        for threads, procs in ((1, 1), (4, 8), (8, 8), (64, 256), (256, 256)):
            for ram in (80 * 100 << 20, 16 << 30, 3000 << 30):
                for window in (0, 1, 21):
                    for nc, zs in ((False, False), (True, False), (False, True)):
                        c, chunk = B.plan(st_size, level=level, threads=threads, processors=procs, ramsize=ram,
                                          window=window, no_compress=nc, zstd=zs)
                        p = O.Params()
                        O.lib().lrzo_params_default(ctypes.byref(p))
                        p.compression_level, p.threads, p.processors, p.ramsize = level, threads, procs, ram
                        p.window, p.no_compress, p.zstd = window, int(nc), int(zs)
                        fs = O.FileStats()
                        O.lib().lrzo_plan(ctypes.byref(p), st_size, ctypes.byref(fs))
                        key = (st_size, level, threads, procs, ram, window, nc, zs)
                        assert c.stream_bufsize == fs.stream_bufsize, key
                        assert c.threads_used == fs.threads_used and c.dictSize_used == fs.dict_size, key


@pytest.mark.parametrize("st_size", [0, 5000, 10 << 20, 700 << 20, 16 << 30])
def test_plan_equals_oracle_sizing_stdin_stdout(B, O, st_size):
    """The same with FLAG_STDIN (blocks sized from the first max_mmap-byte chunk, src/rzip.c:1014-1017, 1075;
    mmap_stdin 835) and FLAG_STDOUT (maxram = ramsize / 6, src/util.c:179-188)."""
    import ctypes
    for level in (5, 7):
        for threads, procs in ((1, 1), (8, 8), (256, 256)):
            for ram in (80 * 100 << 20, 16 << 30, 3000 << 30):
                for window in (0, 1, 21):
                    for si, so in ((1, 0), (0, 1), (1, 1)):
                        c, chunk = B.plan(st_size, level=level, threads=threads, processors=procs, ramsize=ram,
                                          window=window, stdin_mode=si, stdout_mode=so)
                        p = O.Params()
                        O.lib().lrzo_params_default(ctypes.byref(p))
                        p.compression_level, p.threads, p.processors, p.ramsize = level, threads, procs, ram
                        p.window, p.stdin_mode, p.stdout_mode = window, si, so
                        fs = O.FileStats()
                        O.lib().lrzo_plan(ctypes.byref(p), st_size, ctypes.byref(fs))
                        key = (st_size, level, threads, procs, ram, window, si, so)
                        assert c.stream_bufsize == fs.stream_bufsize, key
                        assert c.threads_used == fs.threads_used and c.dictSize_used == fs.dict_size, key
                        assert c.backoff_would_apply in (0, 1)


def test_plan_reports_the_malloc_backoff(B):
    """open_stream_out() probes malloc(limit + overhead x threads) and shrinks `limit` while it fails
    (src/stream.c:1290-1305); by default the library sizes the blocks as if the first probe succeeded and says when this
    host would have refused it."""
    c, _ = B.plan(1 << 40, level=9, threads=256, processors=256, ramsize=1 << 60)  # an absurd -m: no host gives that
    assert c.backoff_would_apply == 1
    c, _ = B.plan(10 << 20, level=7, threads=1, processors=1, ramsize=80 * 100 << 20)
    assert c.backoff_would_apply == 0


def test_malloc_backoff_under_an_address_space_limit(B, O):
    """control->malloc_probe = 1, the probe for real: in a child process whose address space is capped (RLIMIT_AS) so that
    the first probe of the plan is refused, the library and the oracle take the same tenths off `limit` and size the
    blocks alike; without the cap the same plan takes no step."""
    import subprocess
    import textwrap
    code = textwrap.dedent("""
        import ctypes, os, resource, sys
        sys.path.insert(0, %r)
        from conftest import load_bindings
        import oracle_lib as O
        B = load_bindings()
        O.build()
        def both(cap):
            if cap:
                with open('/proc/self/statm') as f:
                    vm = int(f.read().split()[0]) * os.sysconf('SC_PAGE_SIZE')
                resource.setrlimit(resource.RLIMIT_AS, (vm + cap, vm + cap))
            c, chunk = B.plan(4 << 30, level=7, threads=4, processors=8, ramsize=12 << 30, malloc_probe=True)
            p = O.Params(); O.lib().lrzo_params_default(ctypes.byref(p))
            p.compression_level, p.threads, p.processors, p.ramsize, p.malloc_probe = 7, 4, 8, 12 << 30, 1
            fs = O.FileStats(); rc = O.lib().lrzo_plan(ctypes.byref(p), 4 << 30, ctypes.byref(fs))
            return c.backoff_would_apply, c.stream_bufsize, fs.stream_bufsize, c.threads_used, fs.threads_used
        free = both(0)
        capped = both(3 << 30)
        print(free, capped)
        assert free[0] == 0 and free[1] == free[2]
        assert capped[0] >= 1 and capped[1] == capped[2] and capped[3] == capped[4]
        assert capped[1] <= free[1]
    """ % os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr


def test_container_store_equals_oracle_no_compress(B, O):
    """Block order, headers, chaining, magic: product container writer vs oracle for -n, 1 and 3 chunks."""
    data = datagen.long_range(3 * 1048576 + 17, seed=2) + datagen.random_bytes(1500000, seed=3)
    for window_bytes in (0, 2 << 20):
        # emulate -w with a chunk size below 100 MiB via ramsize: max_chunk = ramsize/3*2
        ram = 80 * 100 << 20 if not window_bytes else window_bytes * 3 // 2
        want, fs = O.compress_buffer(data, no_compress=1, threads=1, ramsize=ram)
        c, chunk = B.plan(len(data), no_compress=True, threads=1, ramsize=ram)
        sizes, s0s, s1s = [], [], []
        off, vr = 0, 0
        while off < len(data) or not sizes:
            n = min(chunk, len(data) - off)
            s0, s1, st, crc, vr = O.rzip_chunk(data[off:off + n], level=7, chunk_bytes=B.chunk_bytes_for(n), victim_round=vr)
            sizes.append(n); s0s.append(s0); s1s.append(s1)
            off += n
        assert len(sizes) == fs.n_chunks
        got = B.container_store(len(data), sizes, s0s, s1s, hashlib.md5(data).digest(), no_compress=True, threads=1, ramsize=ram)
        assert got == want


def test_container_store_tiny_inputs(B, O):
    for n in (0, 1, 30, 31, 100):
        data = datagen.text_like(n, seed=4)
        want, _ = O.compress_buffer(data, no_compress=1, threads=1)
        s0, s1, st, crc, vr = O.rzip_chunk(data, level=7)
        got = B.container_store(n, [n], [s0], [s1], hashlib.md5(data).digest(), no_compress=True, threads=1)
        assert got == want, n


def test_hash_index_frozen_table(B, O):
    a = (C.c_uint64 * 256)()
    B.lib().lrzgpu_hash_index(a)
    assert list(a) == O.hash_index()


def test_ctypes_mirrors_match_the_header(B, tmp_path):
    """The Python mirrors of the ABI structs (bindings.py, bench.py) have the sizes and field offsets the
    C header gives them (compiled here with gcc)."""
    import ctypes as C
    import importlib.util
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "sz.c"
    src.write_text('''#include <stdio.h>
#include <stddef.h>
#include "lrzgpu.h"
int main(void) {
  printf("%zu %zu %zu %zu\\n", sizeof(lrzgpu_control), sizeof(lrzgpu_profile), sizeof(lrzgpu_info), sizeof(lrzgpu_scan_stats));
  printf("%zu %zu %zu %zu %zu\\n", offsetof(lrzgpu_control, ramsize), offsetof(lrzgpu_control, st_size),
         offsetof(lrzgpu_control, stream_bufsize), offsetof(lrzgpu_control, zstd_level), offsetof(lrzgpu_profile, resolve_dbg));
  printf("%zu %zu\\n", offsetof(lrzgpu_profile, spec_cancelled_blocks), offsetof(lrzgpu_info, stream_u_len));
  return 0; }''')
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-I", os.path.join(root, "include"), str(src), "-o", str(exe)])
    lines = subprocess.check_output([str(exe)]).decode().split("\n")
    spec = importlib.util.spec_from_file_location("lrz_bench_abi", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    sizes = [int(x) for x in lines[0].split()]
    assert sizes == [C.sizeof(B.Control), C.sizeof(bench.Profile), C.sizeof(B.Info), C.sizeof(B.ScanStats)]
    offs = [int(x) for x in lines[1].split()]
    assert offs == [B.Control.ramsize.offset, B.Control.st_size.offset, B.Control.stream_bufsize.offset,
                    B.Control.zstd_level.offset, bench.Profile.resolve_dbg.offset]
    offs = [int(x) for x in lines[2].split()]
    assert offs == [bench.Profile.spec_cancelled_blocks.offset, B.Info.stream_u_len.offset]


@pytest.mark.parametrize("level,kind", [(7, "text"), (7, "longrange"), (5, "phrases"), (9, "text"), (7, "zeros"), (6, "sparse")])
def test_parser_list_formats(B, O, level, kind):
    """The host parser reads the finder's three list formats (plain couples, tail flag in bit 31, one packed word
    per pair): each must write the reference LzmaCompress bytes.  Lists and tail flags come from the oracle finder / numpy here (no GPU)."""
    import numpy as np
    n = 400000 if kind != "zeros" else 90000
    data = datagen.KINDS[kind](n, seed=17)
    fb = 32 if level < 7 else 64
    dict_size = {5: 1 << 24, 6: 1 << 25, 7: 1 << 25, 9: 1 << 27}[level]
    offs, pairs = O.mf_bt4(data, dict_size=dict_size, fb=fb, cut=16 + fb // 2)
    counts = np.diff(offs).astype(np.uint8)
    rc, want, _ = O.lzma_compress_ref(data, level=level, dict_size=dict_size)
    assert rc == 0
    for fmt in (0, 1, 2):
        if fmt == 2 and dict_size > (1 << 25):
            continue
        lists = B.format_lists(data, counts, pairs, fmt)
        rc, got = B.lzma_encode_with_lists(data, counts, lists, level=level, dict_size=dict_size, fb=fb, list_format=fmt)
        assert rc == 0 and got == want, fmt  # (the AVX2 build on an AVX-512 host: the fresh-process test below)


def test_parser_avx2_build_in_a_fresh_process(B, O):
    """LRZGPU_NO_AVX512=1 selects the AVX2 build of the parser at first use: run it in its own interpreter."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import sys, os
sys.path.insert(0, os.path.join(%r, "tests"))
import numpy as np
import datagen, oracle_lib as O
from conftest import load_bindings
B = load_bindings()
data = datagen.text_like(300000, seed=23) + datagen.long_range(200000, seed=24)
offs, pairs = O.mf_bt4(data, dict_size=1 << 25, fb=64, cut=48)
counts = np.diff(offs).astype(np.uint8)
rc, want, _ = O.lzma_compress_ref(data, level=7, dict_size=1 << 25)
for fmt in (0, 1, 2):
    rc, got = B.lzma_encode_with_lists(data, counts, B.format_lists(data, counts, pairs, fmt), level=7, dict_size=1 << 25, fb=64, list_format=fmt)
    assert rc == 0 and got == want, fmt
print("ok")
''' % root
    env = dict(os.environ, LRZGPU_NO_AVX512="1")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stderr[-2000:]


def test_bench_pmc_traffic_is_tied_to_the_kernels_code_object(monkeypatch, tmp_path):
    """bench.py reports roofline.traffic from the committed PMC summary only for the device code it was measured on:
    the same gfx950 code objects (.hip_fatbin of csrc/*.hip.o), or another build in which THAT kernel's unit has the
    same code object.  Host-only edits -- a header only host code reads, the host half of a .hip file's text -- do not
    invalidate a summary (round 3 lost its last PMC pass to exactly that)."""
    import importlib.util, json, shutil
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_for_test", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    ids = bench.unit_build_ids()
    assert set(bench.KERNEL_UNIT.values()) <= set(ids) and all(len(v) == 16 for v in ids.values())
    # a scratch tree: the device sources, their objects + a summary written for them
    src = os.path.join(root, "lrzip-next_amd", "csrc")
    dst = tmp_path / "lrzip-next_amd" / "csrc"
    dst.mkdir(parents=True)
    for f in os.listdir(src):
        if f.endswith((".hip", ".h", ".hip.o")):
            shutil.copy(os.path.join(src, f), dst / f)
    assert (dst / "rzip_scan.hip.o").exists(), "build() first: the identity is read from the objects"
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    assert bench.build_id() and bench.unit_build_ids() == ids
    (tmp_path / "profiles").mkdir()
    summary = {"build_id": bench.build_id(), "unit_build_ids": bench.unit_build_ids(), "workload_key": "w",
               "kernels": {"k_resolve_mw": {"bytes_per_launch": 123.0}, "k_bt": {"bytes_per_launch": 456.0}}, "note": "n"}
    (tmp_path / "profiles" / "pmc_summary.json").write_text(json.dumps(summary))
    assert bench.pmc_traffic("k_resolve", "w") == (123, "n")
    assert bench.pmc_traffic("k_resolve", "other")[0] is None
    # source text and headers change, the objects do not (host-only edits): everything stands
    for name in ("lzma_mf.hip", "rzip_resolve_mw.h", "lzma_enc.h"):
        with open(dst / name, "a") as f:
            f.write("// changed\n")
    assert bench.pmc_traffic("k_resolve", "w") == (123, "n") and bench.pmc_traffic("k_bt", "w")[0] == 456

    def flip_device_code(name):
        obj = dst / name
        fat = bench._elf_section(str(obj), ".hip_fatbin")
        raw = bytearray(obj.read_bytes())
        at = raw.find(fat)
        raw[at + len(fat) // 2] ^= 0xFF
        obj.write_bytes(bytes(raw))

    # the finder's code object changes: the resolver's figure stands (and says so), the finder's does not
    flip_device_code("lzma_mf.hip.o")
    got, note = bench.pmc_traffic("k_resolve", "w")
    assert got == 123 and "rzip_scan.hip" in note and "unchanged" in note
    assert bench.pmc_traffic("k_bt", "w")[0] is None
    # the resolver's changes: nothing stands
    flip_device_code("rzip_scan.hip.o")
    assert bench.pmc_traffic("k_resolve", "w")[0] is None


@pytest.mark.parametrize("level,kind", [(7, "text"), (7, "longrange"), (7, "zeros"), (7, "few"), (5, "phrases"), (3, "text")])
def test_parser_started_on_a_prefix_of_the_lists(B, O, level, kind):
    """Early start of a block: the parser begins on the lists of the first positions and asks for
    the rest when its search could reach beyond them.  The harness behind lrzgpu_lzma_encode_with_lists_staged hides
    everything from early_positions on (bytes overwritten, lists cut off) until then; the stream must be the
    reference's LzmaCompress bytes whatever the switch position -- inside a literal run, inside a long match, inside
    the first search window, at the very start or beyond the end."""
    import numpy as np
    n = 300000 if kind != "zeros" else 90000
    data = datagen.KINDS[kind](n, seed=23)
    fb = 32 if level < 7 else 64
    dict_size = {3: 1 << 20, 5: 1 << 24, 7: 1 << 25}[level]
    if level >= 5:
        offs, pairs = O.mf_bt4(data, dict_size=dict_size, fb=fb, cut=16 + fb // 2)
    else:
        offs, pairs = O.mf_hc5(data, dict_size=dict_size, fb=fb, cut=(16 + fb // 2) // 2)
    counts = np.diff(offs).astype(np.uint8)
    rc, want, _ = O.lzma_compress_ref(data, level=level, dict_size=dict_size)
    assert rc == 0
    for fmt in ((0, 2) if level >= 5 else (0,)):
        lists = B.format_lists(data, counts, pairs, fmt)
        for early in (0, 1, 1000, 2658, 2659, 5000, 77777, n // 2 + 3, n - 3000, n - 1, n, n + 10):
            rc, got = B.lzma_encode_with_lists_staged(data, counts, lists, early, level=level, dict_size=dict_size, fb=fb, list_format=fmt)
            assert rc == 0 and got == want, (fmt, early)
        # many stages (the whole-file driver hands a block over piece by piece while its scan runs): steps shorter than
        # the parser's reach (several requests before one search), of the reach, and long ones
        for early, step in ((0, 1), (1000, 700), (5000, 2659), (3, 4096), (20000, 50001), (n // 2, 65536)):
            rc, got = B.lzma_encode_with_lists_staged(data, counts, lists, early, level=level, dict_size=dict_size, fb=fb, list_format=fmt,
                                                      stage_step=step)
            assert rc == 0 and got == want, (fmt, early, step)


@pytest.mark.parametrize("kind", ["text", "few", "phrases"])
def test_parser_started_on_the_lists_of_a_prefix_run(B, O, kind):
    """The early start end to end on the host: the early lists come from a finder that saw only a PREFIX of the block
    (the restated reference finder on data[:P], tail flags computed on that prefix too), are used below P - fb - 4, and
    the whole block's lists take over from there.  Reference LzmaCompress bytes.  (The dictionary is no larger than
    the prefix so that the oracle finder derives the block's hash mask for it; the GPU finder takes the block size as
    a parameter.)"""
    import numpy as np
    n, fb, dict_size, level = 1200000, 64, 1 << 19, 7
    data = datagen.KINDS[kind](n, seed=29)
    offs, pairs = O.mf_bt4(data, dict_size=dict_size, fb=fb, cut=16 + fb // 2)
    counts = np.diff(offs).astype(np.uint8)
    rc, want, _ = O.lzma_compress_ref(data, level=level, dict_size=dict_size)
    assert rc == 0
    for P in (700000, 1048576 + 5):
        poffs, ppairs = O.mf_bt4(data[:P], dict_size=dict_size, fb=fb, cut=16 + fb // 2)
        pcounts = np.diff(poffs).astype(np.uint8)
        for fmt in (0, 2):
            lists = B.format_lists(data, counts, pairs, fmt)
            plists = B.format_lists(data[:P], pcounts, ppairs, fmt)
            rc, got = B.lzma_encode_with_lists_staged(data, counts, lists, P - fb - 4, level=level, dict_size=dict_size, fb=fb,
                                                      list_format=fmt, early_counts=pcounts, early_pairs=plists)
            assert rc == 0 and got == want, (P, fmt)


def test_rccl_transport_entry_points_check_their_arguments(B):
    """csrc/shard_rccl.cpp on a box without a GPU: the library loads (RCCL is dlopen'd, not linked), the entry points
    exist and refuse bad arguments before touching a device."""
    import ctypes as C
    L = B.lib()
    assert L.lrzgpu_rccl_available() in (0, 1)
    assert L.lrzgpu_rccl_unique_id(None) == -101  # LRZGPU_E_PARAM
    raw = (C.c_ubyte * 128)()
    assert L.lrzgpu_rccl_comm_create(raw, 2, 2, 0, None) == -101
    assert L.lrzgpu_rccl_comm_create(raw, 2, 2, 0, raw) == -101  # rank outside the world
    assert L.lrzgpu_rccl_comm_destroy(None) == -101
    assert L.lrzgpu_rccl_loopback(None, None, None, 0) == -101
