#!/bin/bash
# On the GPU box: the headline step with the library as built (B) and with tools/_bin/liblrzgpu_r4parser.so (A: the same
# library with round 4's parser objects), alternating, same box, same minutes -- what the parser work is worth in the
# pipeline (boxes differ by ~5 % in host speed, more than the effect measured).
cd "$(dirname "$0")/../.."
L=lrzip-next_amd/liblrzgpu.so
cp $L /tmp/lib_B.so; cp tools/_bin/liblrzgpu_r4parser.so /tmp/lib_A.so
for round in 1 2; do
  for v in A B; do
    cp /tmp/lib_$v.so $L
    python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-file-leg > /tmp/ab_$v$round.json 2>/dev/null
    python - <<PY
import json
d=json.load(open('/tmp/ab_$v$round.json'))
print("$v round $round: %.1f MB/s  %.0f ms/step  encoders %.1f CPU-s/step  busy %.1f idle %.1f" % (d['value'], d['ms_per_step'], d['config']['host_cpu_seconds_per_step_by_thread_role']['encoders (parser + range coder)'], d['critical_path']['encoders_busy_s_per_step'], d['critical_path']['encoders_idle_s_per_step']))
PY
  done
done
cp /tmp/lib_B.so $L
