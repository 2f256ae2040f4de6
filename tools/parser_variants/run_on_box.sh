#!/bin/bash
# On the GPU box (EPYC 9575F): lists of a 64 MiB block of the bench text from the GPU finder, then every harness binary
# under tools/_bin/pv_* with 1 and 16 threads (+ hardware counters where allowed), the *_prof ones once for their laps.
cd "$(dirname "$0")/../.."
OUT=gpurun_out/parser_variants; mkdir -p $OUT
python tools/parser_variants/make_input.py ${MIB:-64} /tmp/pv 2>&1 | tail -1 | tee $OUT/log.txt
for b in tools/_bin/pv_*; do
  case $b in *_prof) continue;; esac
  echo "== $(basename $b)" | tee -a $OUT/log.txt
  $b /tmp/pv 1 2 2>&1 | tee -a $OUT/log.txt
  $b /tmp/pv 16 2 2>&1 | tee -a $OUT/log.txt
done
for b in tools/_bin/pv_*_prof; do
  echo "== $(basename $b): cycle laps per section, one thread" | tee -a $OUT/laps.txt
  $b /tmp/pv 1 1 2>&1 | tee -a $OUT/laps.txt
done
