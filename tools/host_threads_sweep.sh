#!/bin/bash
# On the GPU box: the headline step with more encoder threads than the CPU quota has CPUs (the quota is a budget of CPU
# time per 100 ms, not a set of cores: a 17th thread uses what the sixteen leave when they wait), same box, same minutes.
cd "$(dirname "$0")/.."
for t in ${@:-16 18 16 20 17 24}; do
  python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-file-leg --host-threads $t > /tmp/ht_$t.json 2>/dev/null
  python - <<PY
import json
d=json.load(open('/tmp/ht_$t.json'))
c=d['critical_path']
print("host_threads %2d: %.1f MB/s  %.0f ms/step  encoders %.1f CPU-s busy %.1f idle %.1f  last encode %.2f s" % ($t, d['value'], d['ms_per_step'], d['config']['host_cpu_seconds_per_step_by_thread_role']['encoders (parser + range coder)'], c['encoders_busy_s_per_step'], c['encoders_idle_s_per_step'], c['last_encode_done_s']))
PY
done
