#!/bin/bash
# Host parser build variants on the GPU box's CPU (EPYC 9575F): tuning flags and profile-guided optimisation.
# Rebuilds only the two lzma_parser objects + relinks, runs tools/parser_gpu_case.py (64 MiB block, T = 1 and 16).
set -e
cd "$(dirname "$0")/.."
C=lrzip-next_amd/csrc
run() { echo "== $1"; python tools/parser_gpu_case.py 64 1,16 2 2>&1 | grep "T=" ; }
build() { # $1 = extra flags
  g++ -O3 -march=x86-64-v3 -std=c++17 -fPIC -Wall -Wno-unused-result $1 -c $C/lzma_parser.cpp -o $C/lzma_parser.host.o
  g++ -O3 -march=x86-64-v4 -std=c++17 -fPIC -Wall -Wno-unused-result $1 -c $C/lzma_parser.cpp -o $C/lzma_parser.v4.host.o
  make -s -C $C >/dev/null
}
run baseline
build "-mtune=znver3"; run "mtune=znver3"
build "-fprofile-generate -fprofile-update=atomic"
sed -i 's/-lpthread -ldl/-lpthread -ldl -lgcov/' $C/Makefile; make -s -C $C > /dev/null
python tools/parser_gpu_case.py 16 1 2 > /dev/null 2>&1   # training run
ls $C/*.gcda
build "-fprofile-use -fprofile-correction -Wno-missing-profile"; run "PGO (trained on a 16 MiB block of the same text)"
build "-fprofile-use -fprofile-correction -Wno-missing-profile -mtune=znver3"; run "PGO + mtune=znver3"
