#!/bin/bash
# Which recorded profile makes the host parser fastest on THIS host (GPU box: EPYC 9575F)?  For every profile given
# (files under lrzip-next_amd/csrc/pgo/, or "none"): rebuild the AVX-512 parser object with it, relink, run
# tools/parser_gpu_case.py (64 MiB block of the bench text, 1 and 16 parser threads, the pipeline's list format).
cd "$(dirname "$0")/.."
C=lrzip-next_amd/csrc
for prof in "$@"; do
  rm -f $C/lzma_parser.v4.host.gcda
  FL=""
  if [ "$prof" != none ]; then cp $C/pgo/$prof $C/lzma_parser.v4.host.gcda; FL="-fprofile-use -fprofile-correction -Wno-missing-profile -Wno-coverage-mismatch"; fi
  g++ -O3 -march=x86-64-v4 -std=c++17 -fPIC -Wall -Wno-unused-result $FL -c $C/lzma_parser.cpp -o $C/lzma_parser.v4.host.o
  rm -f $C/lzma_parser.v4.host.gcda
  (cd $C && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../liblrzgpu.so *.hip.o api_backend.o api_rzip.o driver.o shard.o shard_rccl.o stream_api.o unrzip.o *.host.o -lpthread -ldl)
  echo "== profile: $prof"
  python tools/parser_gpu_case.py 64 1,16 2 2>&1 | grep "T="
done
