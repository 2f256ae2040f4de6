#!/bin/bash
# Host parser build variants on the GPU box's CPU (EPYC 9575F): profile-guided optimisation (and, measured once:
# -mtune=znver3 is 3 % SLOWER than the generic tuning with g++ 11.4).
# Rebuilds only the two lzma_parser objects + relinks, runs tools/parser_gpu_case.py (64 MiB block, T = 1 and 16),
# leaves the recorded profile under gpurun_out/pgo/ (copy lzma_parser.v4.host.gcda to lrzip-next_amd/csrc/pgo/).
set -e
cd "$(dirname "$0")/.."
C=lrzip-next_amd/csrc
run() { echo "== $1"; python tools/parser_gpu_case.py 64 1,16 2 2>&1 | grep "T=" ; }
build() { # $1 = extra flags.  (Only the AVX-512 object is profiled: it is the one this host runs.)
  g++ -O3 -march=x86-64-v3 -std=c++17 -fPIC -Wall -Wno-unused-result -c $C/lzma_parser.cpp -o $C/lzma_parser.host.o
  g++ -O3 -march=x86-64-v4 -std=c++17 -fPIC -Wall -Wno-unused-result $1 -c $C/lzma_parser.cpp -o $C/lzma_parser.v4.host.o
  touch $C/lzma_parser.host.o $C/lzma_parser.v4.host.o
  make -s -C $C >/dev/null
}
build ""; run "baseline (no profile)"
sed -i 's#-lpthread -ldl#-lpthread -ldl /usr/lib/gcc/x86_64-linux-gnu/11/libgcov.a#' $C/Makefile
build "-fprofile-generate -fprofile-update=atomic"
rm -f $C/*.gcda
python tools/pgo_train.py 2>&1 | tail -2   # the training workload
ls $C/*.gcda
mkdir -p gpurun_out/pgo; cp $C/*.gcda gpurun_out/pgo/
build "-fprofile-use -fprofile-correction -Wno-missing-profile"; run "PGO (trained on a 16 MiB block of the same text)"
