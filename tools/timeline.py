#!/usr/bin/env python3
"""Timeline of one traced run (LRZGPU_TRACE=2 stderr): where blocks wait between scan, gate, finder and encoder."""
import sys, collections
ev = collections.defaultdict(dict)
for l in open(sys.argv[1]):
    if not l.startswith("ev "):
        continue
    _, t, what, _, c, _, s, _, off, _, ln = l.split()[:11]
    if what in ("stage_start", "rest"):
        continue  # (tools/ramp.py reads those)
    ev[(int(c), int(s), int(off))][what] = float(t)
    ev[(int(c), int(s), int(off))]["len"] = int(ln)
blocks = [v for v in ev.values() if "enc_start" in v and "enc_end" in v]
T = max(v.get("enc_end", v.get("gpu_end", 0)) for v in ev.values())
print("blocks %d, encoded %d, run %.1f s" % (len(ev), len(blocks), T))
def avg(k1, k2, sel=blocks):
    d = [v[k2] - v[k1] for v in sel if k1 in v and k2 in v]
    return (sum(d) / len(d), max(d)) if d else (0, 0)
for a, b in (("submit", "gpu_start"), ("gpu_start", "gpu_end"), ("submit", "gate_done"), ("gpu_end", "enc_start"), ("gate_done", "enc_start"), ("enc_start", "enc_end")):
    m, mx = avg(a, b)
    print("  %-10s -> %-10s avg %6.2f s  max %6.2f s" % (a, b, m, mx))
step = 1.0
print("per second: blocks submitted | in gpu worker | waiting for encoder (finder+gate done) | encoding")
t = 0.0
while t < T + step:
    sub = sum(1 for v in ev.values() if t <= v.get("submit", -1) < t + step)
    ingpu = sum(1 for v in ev.values() if v.get("gpu_start", 1e9) <= t < v.get("gpu_end", -1))
    ready = sum(1 for v in blocks if max(v.get("gpu_end", 0), v.get("gate_done", 0)) <= t < v["enc_start"])
    enc = sum(1 for v in blocks if v["enc_start"] <= t < v["enc_end"])
    print("  t=%5.1f  submitted %3d  gpu %2d  ready-waiting %3d  encoding %2d" % (t, sub, ingpu, ready, enc))
    t += step
