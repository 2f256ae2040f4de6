#!/bin/bash
# Same box, two trees: the resolver on the degenerate inputs, on the bench text alone, and inside a bench step (4 GiB of the headline file: long-range matches).
# usage (GPU box): bash tools/ab_resolver.sh <other tree> > gpurun_out/ab_resolver.log
other=${1:-tools/_bin/headtree}
for tree in "$other" .; do
	echo "===== tree $tree ($(cd $tree && git rev-parse --short HEAD 2>/dev/null) $( [ "$tree" = . ] && echo '+ working copy'))"
	timeout 300 python $tree/tools/slow_phrases.py 2>&1 | grep -v amdgpu.ids | grep -v "lists alone"
	timeout 200 python $tree/tools/resolve_prof.py 256 text 2>&1 | grep "^scan"
	(cd $tree && timeout 600 python bench.py --steps 2 --warmup 1 --mib 4096 --base-mib 256 --window 5 --no-cpu-baseline --no-file-leg 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('bench, 4 GiB of the headline file: %.1f %s, %.0f ms per step; k_resolve_mw %.1f ms per launch, k_bt %.0f ms per step; rounds %d exact steps %d stopped at a match %d' % (d['value'], d['unit'], d['ms_per_step'], r.get('avg_launch_ms', -1), r['per_kernel_ms_per_step']['k_bt'], r['resolver']['batches'], r['resolver']['serial_steps'], r['resolver']['stop_match']))")
done
