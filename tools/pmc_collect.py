#!/usr/bin/env python3
"""Run on the GPU box (inside gpurun): rocprofv3 kernel trace + the two HBM PMC passes of the default bench
command, reduced to small summaries under gpurun_out/prof/ (copy what is to be judged into profiles/):
  <tag>_kernel_stats.csv     rocprofv3 --kernel-trace --stats (per-kernel calls / total / average duration)
  <tag>_bench_under_rocprof.json  the bench line of that same run (HIP-event figures to compare with)
  <tag>_pmc_hbm.csv          per kernel: launches, FETCH_SIZE and WRITE_SIZE totals and per launch (raw KB)
  pmc_summary.json           what bench.py reads for roofline.traffic: bytes per launch, tied to the build id
PMC passes are separate runs with no tracing besides --kernel-trace (MI355X_MICROARCH.md, HBM section);
FETCH_SIZE on gfx950 counts 64 B per 128 B request of a wide streaming read, so the summary reports
fetch x 2 + write for kernels marked streaming and raw fetch + write for the others, and says which."""
import csv, glob, json, os, re, shutil, subprocess, sys, collections

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "gpurun_out", "prof")
TAG = sys.argv[1] if len(sys.argv) > 1 else "r2"
BENCH = ["python", os.path.join(ROOT, "bench.py"), "--steps", "1", "--no-cpu-baseline", "--no-file-leg"] + sys.argv[2:]
STREAMING = {"k_crc32_tiles", "k_long_compare", "k_gather_runs", "k_tag_scan"}


def short(name):
    m = re.match(r"(?:void )?(?:lrzgpu::)?([A-Za-z_0-9]+)", name.strip('"'))
    n = m.group(1) if m else name
    if n.startswith("rocprim") or "rocprim" in name[:40]:
        # rocPRIM's kernels by their own names (round 5's summaries had them under one label: hard to audit)
        r = re.search(r"rocprim::(?:ROCPRIM_[0-9]+_NS::)?(?:detail::)?([A-Za-z_0-9]+)", name)
        return "rocprim::" + (r.group(1) if r else "kernel")
    return n


RAW = "/tmp/lrzgpu_prof_raw"  # raw rocprofv3 output is hundreds of MiB: it never goes under gpurun_out/


def run(args, tag, bench_extra=()):
    d = os.path.join(RAW, tag)
    shutil.rmtree(d, ignore_errors=True)
    os.makedirs(d, exist_ok=True)
    env = dict(os.environ, TMPDIR="/tmp")
    p = subprocess.run(["rocprofv3"] + args + ["-d", d, "-o", tag, "--output-format", "csv", "--"] + BENCH + list(bench_extra), cwd="/tmp", env=env,
                       capture_output=True, text=True)
    line = [l for l in p.stdout.splitlines() if l.startswith("{")]
    print("rocprofv3", tag, "rc", p.returncode, "files", sum(len(f) for _, _, f in os.walk(d)), flush=True)
    if p.returncode != 0 or not line:
        sys.stderr.write(p.stdout[-2000:] + p.stderr[-4000:])
        raise SystemExit("rocprofv3 %s failed" % tag)
    return d, json.loads(line[-1])


def main():
    os.makedirs(OUT, exist_ok=True)
    # the kernel trace in the driver's command shape: several timed steps (its launch averages are what the driver's line
    # is compared with); the counter passes take one step each
    d, line = run(["--kernel-trace", "--stats"], TAG + "_kt", ["--steps", "3"])
    json.dump(line, open(os.path.join(OUT, TAG + "_bench_under_rocprof.json"), "w"), indent=1)
    for f in glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True):
        rows = list(csv.reader(open(f)))
        with open(os.path.join(OUT, TAG + "_kernel_stats.csv"), "w", newline="") as o:
            w = csv.writer(o)
            w.writerow(rows[0])
            for r in rows[1:]:
                w.writerow([short(r[0])] + r[1:])
    totals = collections.defaultdict(lambda: collections.defaultdict(float))
    launches = collections.Counter()
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d, _ = run(["--kernel-trace", "--pmc", counter], TAG + "_" + counter.lower())
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                k = short(r.get("Kernel_Name", ""))
                if r.get("Counter_Name") == counter:
                    totals[k][counter] += float(r["Counter_Value"])
                    if counter == "FETCH_SIZE":
                        launches[k] += 1
    with open(os.path.join(OUT, TAG + "_pmc_hbm.csv"), "w", newline="") as o:
        w = csv.writer(o)
        w.writerow(["kernel", "launches", "FETCH_SIZE_KB_total", "WRITE_SIZE_KB_total", "FETCH_SIZE_KB_per_launch", "WRITE_SIZE_KB_per_launch"])
        for k in sorted(totals, key=lambda k: -(totals[k]["FETCH_SIZE"] + totals[k]["WRITE_SIZE"])):
            n = max(launches[k], 1)
            w.writerow([k, launches[k], "%.1f" % totals[k]["FETCH_SIZE"], "%.1f" % totals[k]["WRITE_SIZE"],
                        "%.1f" % (totals[k]["FETCH_SIZE"] / n), "%.1f" % (totals[k]["WRITE_SIZE"] / n)])
    import bench
    kernels = {}
    for k in totals:
        n = max(launches[k], 1)
        f = totals[k]["FETCH_SIZE"] * (2 if k in STREAMING else 1)
        kernels[k] = {"launches": launches[k], "bytes_per_launch": (f + totals[k]["WRITE_SIZE"]) * 1024.0 / n,
                      "fetch_correction": 2 if k in STREAMING else 1}
    # the BT walk is one launch per finder run since round 4 (k_bt_walk); bench.py calls it "k_bt"
    if "k_bt_walk" in kernels:
        kernels["k_bt"] = dict(kernels["k_bt_walk"])
    if "k_bt_wave" in kernels and "k_bt" in kernels:
        n = max(kernels["k_bt"]["launches"], 1)
        kernels["k_bt+k_bt_wave"] = {"launches": kernels["k_bt"]["launches"],
                                     "bytes_per_launch": kernels["k_bt"]["bytes_per_launch"] +
                                     kernels["k_bt_wave"]["bytes_per_launch"] * kernels["k_bt_wave"]["launches"] / n,
                                     "fetch_correction": 1}
    cfg = line["config"]["workload"]
    wk = re.match(r"(cfg\d)", cfg).group(1)
    a = dict(zip(sys.argv[2::2], sys.argv[3::2]))
    mib = int(a.get("--mib", 16384 if wk == "cfg3" else 4096))
    window = int(a.get("--window", 21 if wk == "cfg3" else 0))
    json.dump({"build_id": bench.build_id(), "unit_build_ids": bench.unit_build_ids(), "workload_key": "%s-%dMiB-w%d-%s" % (wk, mib, window, a.get("--alphabet", "alnum")),
               "kernels": kernels,
               "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), KB x 1024 per launch; fetch x 2 for the wide "
                       "streaming kernels (gfx950 counts 64 B per 128 B request), raw for the narrow random-access ones (uncalibrated: "
                       "true reads are between 1x and 2x the fetch part)"},
              open(os.path.join(OUT, "pmc_summary.json"), "w"), indent=1)
    shutil.rmtree(RAW, ignore_errors=True)
    print("wrote", OUT, os.listdir(OUT))


main()
