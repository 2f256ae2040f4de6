#!/bin/bash
# Host parser build flags on THIS host (GPU box: EPYC 9575F): the AVX-512 object rebuilt with the committed profile plus
# each extra flag set given (one argument per set, "" = the Makefile's), relinked, timed with tools/parser_gpu_case.py
# (64 MiB block of the bench text, 16 parser threads twice, the pipeline's list format).
cd "$(dirname "$0")/.."
C=lrzip-next_amd/csrc
cp $C/../liblrzgpu.so /tmp/liblrzgpu_keep.so
for fl in "$@"; do
  cp $C/pgo/lzma_parser.v4.host.gcda $C/lzma_parser.v4.host.gcda
  g++ -O3 -march=x86-64-v4 -std=c++17 -fPIC -Wall -Wno-unused-result -fprofile-use -fprofile-correction -Wno-missing-profile -Wno-coverage-mismatch $fl -c $C/lzma_parser.cpp -o $C/lzma_parser.v4.host.o 2>&1 | head -3
  rm -f $C/lzma_parser.v4.host.gcda
  (cd $C && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../liblrzgpu.so *.hip.o api_backend.o api_rzip.o driver.o shard.o shard_rccl.o stream_api.o unrzip.o *.host.o -lpthread -ldl)
  echo "== flags: '$fl'"
  python tools/parser_gpu_case.py 64 16,16 2 2>&1 | grep "T="
done
cp /tmp/liblrzgpu_keep.so $C/../liblrzgpu.so
