#!/usr/bin/env python3
"""Run on the GPU box (inside gpurun): the LDS-resident bucket kernels of the BT4 finder (LRZGPU_BT_LDS_MIN) --
  1. the parity test of those launches (tests/test_backend_gpu.py -k lds_resident),
  2. tools/bt_case.py on one 64 MiB block of the bench text: time per setting, lists compared with the default's,
  3. HBM traffic of the finder's kernels on that block (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes,
     --kernel-trace only beside them) for the default and for the settings named on the command line.
Everything is bounded by its own timeout; summaries go to gpurun_out/btlds/."""
import collections, csv, glob, os, re, shutil, subprocess, sys, time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out", "btlds")
RAW = "/tmp/lrzgpu_btlds_raw"
os.makedirs(OUT, exist_ok=True)
T0 = time.time()


def sh(cmd, log, limit, env=None, cwd=ROOT):
    t = time.time()
    try:
        p = subprocess.run(cmd, cwd=cwd, env=env, capture_output=True, text=True, timeout=limit)
        rc, out = p.returncode, p.stdout + p.stderr
    except subprocess.TimeoutExpired as e:
        rc, out = -9, (e.stdout or b"").decode(errors="replace") + (e.stderr or b"").decode(errors="replace") + "\nTIMEOUT\n"
    open(os.path.join(OUT, log), "w").write(out)
    print("[%5.0f s] %s: rc %d in %.0f s" % (time.time() - T0, log, rc, time.time() - t), flush=True)
    print(out[-1500:], flush=True)
    return rc


def short(name):
    name = name.strip('"')
    m = re.search(r"(k_bt_wave<\d+u?>|k_bt_wave|k_[a-z0-9_]+)", name)
    if m:
        return m.group(1)
    return "rocprim_sort/scan/select" if "rocprim" in name else name[:40]


def pmc(setting, tag):
    totals = collections.defaultdict(lambda: collections.defaultdict(float))
    launches = collections.Counter()
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = os.path.join(RAW, tag + "_" + counter)
        shutil.rmtree(d, ignore_errors=True)
        os.makedirs(d, exist_ok=True)
        rc = sh(["rocprofv3", "--kernel-trace", "--pmc", counter, "-d", d, "-o", "x", "--output-format", "csv", "--",
                 "python", os.path.join(ROOT, "tools", "bt_case.py"), "64", setting], "pmc_%s_%s.log" % (tag, counter), 150,
                env=dict(os.environ, TMPDIR="/tmp", BT_CASE_REPS="1"), cwd="/tmp")
        if rc:
            return
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if r.get("Counter_Name") == counter:
                    k = short(r.get("Kernel_Name", ""))
                    totals[k][counter] += float(r["Counter_Value"])
                    if counter == "FETCH_SIZE":
                        launches[k] += 1
    with open(os.path.join(OUT, "pmc_hbm_%s.csv" % tag), "w", newline="") as o:
        w = csv.writer(o)
        w.writerow(["kernel", "launches", "FETCH_SIZE_KB_total", "WRITE_SIZE_KB_total"])
        for k in sorted(totals, key=lambda k: -(totals[k]["FETCH_SIZE"] + totals[k]["WRITE_SIZE"])):
            w.writerow([k, launches[k], "%.1f" % totals[k]["FETCH_SIZE"], "%.1f" % totals[k]["WRITE_SIZE"]])
    print(open(os.path.join(OUT, "pmc_hbm_%s.csv" % tag)).read(), flush=True)
    shutil.rmtree(RAW, ignore_errors=True)


def main():
    settings = sys.argv[1:] or ["4096,256", "4096,1024"]
    sh(["python", "-m", "pytest", "tests/test_backend_gpu.py", "-x", "-q", "-k", "lds_resident", "-p", "no:cacheprovider"], "pytest.log", 240)
    sh(["python", "tools/bt_case.py", "64", "4096"] + settings + ["4096,2048"], "bt_case.log", 200)
    pmc("4096", "default")
    pmc(settings[0], "lds" + settings[0].split(",")[1])


main()
