#!/bin/bash
mkdir -p gpurun_out/r2e
exec > gpurun_out/r2e/log.txt 2>&1
set -x
cd /tmp && export TMPDIR=/tmp
timeout 2400 python $GRAFT_REPO_ROOT/tools/pmc_collect.py r2 2>&1 | tail -40
cd $GRAFT_REPO_ROOT
ls -la gpurun_out/prof; du -sh gpurun_out
timeout 1200 python -m pytest tests/test_configs_gpu.py -x -q -m gpu 2>&1 | tail -8
