#!/bin/bash
# Host-side NUMA placement experiment on the GPU box (2 sockets, 16-CPU CFS quota, no cpuset): the headline step with
# the scheduler's own placement, then with the whole process confined to the CPUs of one memory node (first-touch then
# puts the encoders' block bytes, lists and search windows on that node too).  One warm-up step + one timed step each.
mkdir -p gpurun_out/numa
O=gpurun_out/numa
{
  echo "allowed: $(grep Cpus_allowed_list /proc/self/status)"
  for n in /sys/devices/system/node/node*; do echo "$(basename $n): cpus $(cat $n/cpulist)  $(grep MemTotal $n/meminfo | awk '{print $4, $5}')"; done
  for d in /sys/class/drm/card*/device; do [ -f $d/numa_node ] && echo "$d numa_node $(cat $d/numa_node) vendor $(cat $d/vendor)"; done
  cat /sys/fs/cgroup/cpu.max 2>/dev/null
} > $O/topology.txt 2>&1
cat $O/topology.txt
run() { name=$1; shift; echo "== $name"; timeout 170 "$@" python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-file-leg > $O/$name.json 2> $O/$name.err; python - $O/$name.json <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print("  %.1f MB/s  %.0f ms/step  encoders %s CPU-s" % (d["value"], d["ms_per_step"], d["config"]["host_cpu_seconds_per_step_by_thread_role"].get("encoders (parser + range coder)")))
except Exception as e:
    print("  no line:", e)
PY
}
run default env X=1
# the node of the (first) AMD GPU, and one other node
G=-1
for d in /sys/class/drm/card*/device; do
  if [ -f $d/numa_node ] && [ "$(cat $d/vendor)" = "0x1002" ]; then G=$(cat $d/numa_node); break; fi
done
[ "$G" -lt 0 ] && G=0
OTHER=""
for n in /sys/devices/system/node/node*; do k=${n##*node}; if [ "$k" != "$G" ]; then OTHER=$k; break; fi; done
echo "gpu node $G, other node $OTHER"
run gpu_node$G taskset -c "$(cat /sys/devices/system/node/node$G/cpulist)"
[ -n "$OTHER" ] && run other_node$OTHER taskset -c "$(cat /sys/devices/system/node/node$OTHER/cpulist)"
