#!/bin/bash
# What the GPU box's host side looks like: CPU topology, cgroup quota / cpuset, clocks.
mkdir -p gpurun_out
{
echo "== nproc"; nproc; nproc --all
echo "== lscpu"; lscpu | grep -v -i "flags\|vulnerab"
echo "== cgroup"; cat /sys/fs/cgroup/cpu.max 2>/dev/null; cat /sys/fs/cgroup/cpuset.cpus.effective 2>/dev/null; cat /sys/fs/cgroup/cpu/cpu.cfs_quota_us /sys/fs/cgroup/cpu/cpu.cfs_period_us 2>/dev/null
echo "== affinity"; taskset -p $$; grep -i "cpus_allowed_list\|mems_allowed_list" /proc/self/status
echo "== mem"; free -g; cat /sys/fs/cgroup/memory.max 2>/dev/null
echo "== numa"; ls /sys/devices/system/node/ | head; cat /sys/devices/system/node/node*/cpulist 2>/dev/null
echo "== L3 domains"; for c in 0 1 8 16 64 127; do echo -n "cpu$c: "; cat /sys/devices/system/cpu/cpu$c/cache/index3/shared_cpu_list 2>/dev/null; cat /sys/devices/system/cpu/cpu$c/topology/thread_siblings_list 2>/dev/null; done
echo "== rocm-smi"; rocm-smi --showclocks --showpower --showperflevel 2>&1 | head -40
echo "== tmpfs"; df -h /dev/shm /tmp | cat
} > gpurun_out/probe_box.txt 2>&1
