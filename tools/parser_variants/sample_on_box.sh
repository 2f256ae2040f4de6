#!/bin/bash
# On the GPU box: where the parser's cycles / instructions / branch misses are, line by line (harness.cpp PV_SAMPLE).
cd "$(dirname "$0")/../.."
OUT=gpurun_out/parser_sample; mkdir -p $OUT
B=${1:-tools/_bin/pv_current_pgo}
[ -f /tmp/pv/d.bin ] || python tools/parser_variants/make_input.py ${MIB:-64} /tmp/pv 2>&1 | tail -1
for ev in ${EVENTS:-cycles instructions branch-misses l1d-misses cache-misses}; do
  per=200003; [ $ev = branch-misses ] && per=1009; [ $ev = l1d-misses ] && per=2003; [ $ev = cache-misses ] && per=503
  PV_SAMPLE=$ev PV_PERIOD=$per PV_SAMPLE_OUT=/tmp/pv_$ev.txt $B /tmp/pv 1 1 2>&1 | tail -3
  python tools/parser_variants/lines.py $B /tmp/pv_$ev.txt 0.3 > $OUT/$(basename $B)_$ev.txt; cp /tmp/pv_$ev.txt $OUT/$(basename $B)_$ev.raw
done
