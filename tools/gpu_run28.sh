#!/bin/bash
# final measurements of the round: kernel trace + HBM PMC passes of the headline step, then the headline line itself
mkdir -p gpurun_out/r3f
exec > gpurun_out/r3f/log.txt 2>&1
set -x
cd /tmp && export TMPDIR=/tmp
timeout 540 python $GRAFT_REPO_ROOT/tools/pmc_collect.py r2b 2>&1 | tail -4
cd $GRAFT_REPO_ROOT
cp gpurun_out/prof/pmc_summary.json profiles/pmc_summary.json
LRZGPU_TRACE=1 timeout 400 python bench.py --steps 3 --warmup 1 --verify 2> gpurun_out/r3f/bench16g.err > gpurun_out/r3f/bench16g.json
cut -c1-200 gpurun_out/r3f/bench16g.json
grep "lrzgpu driver" gpurun_out/r3f/bench16g.err | cut -c1-330
grep -v "lrzgpu scan\|lrzgpu driver\|^ev " gpurun_out/r3f/bench16g.err | tail -5 | cut -c1-300
rm -f gpurun_out/r3f/bench16g.err
