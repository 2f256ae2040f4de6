#!/bin/bash
mkdir -p gpurun_out/r2n
exec > gpurun_out/r2n/log.txt 2>&1
set -x
cd /tmp && export TMPDIR=/tmp
timeout 2400 python $GRAFT_REPO_ROOT/tools/pmc_collect.py r2 2>&1 | tail -4
cd $GRAFT_REPO_ROOT
cp gpurun_out/prof/pmc_summary.json profiles/pmc_summary.json
LRZGPU_TRACE=1 timeout 1500 python bench.py --steps 5 --warmup 2 --verify 2> gpurun_out/r2n/bench16g.err > gpurun_out/r2n/bench16g.json
cut -c1-200 gpurun_out/r2n/bench16g.json
grep "lrzgpu driver" gpurun_out/r2n/bench16g.err | cut -c1-300
timeout 900 python bench.py --workload cfg2 --steps 2 --warmup 1 --verify 2>/dev/null > gpurun_out/r2n/bench_cfg2.json
cut -c1-200 gpurun_out/r2n/bench_cfg2.json
