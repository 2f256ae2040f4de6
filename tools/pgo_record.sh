#!/bin/bash
# Records the profile of the host parser's AVX-512 object for csrc/Makefile's -fprofile-use build, in a scratch copy of the
# repository (the working tree's objects are not touched): instrumented lzma_parser.v4.host.o, tools/pgo_train.py, the
# .gcda copied to lrzip-next_amd/csrc/pgo/.  Needs an AVX-512 host (the profiled object is the one such hosts run); lists
# come from the GPU finder if a GPU is visible, else from the oracle finder.  Re-run after any change to lzma_parser.cpp,
# lzma_model.h or lzma_rangecoder.h: functions whose source changed lose their profile.
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
W=${PGO_WORK:-/tmp/lrzgpu_pgo}
rm -rf "$W"; mkdir -p "$W"
(cd "$ROOT" && tar cf - --exclude=.git --exclude=gpurun_out --exclude='*.gcda' .) | tar xf - -C "$W"
cd "$W"
C=lrzip-next_amd/csrc
rm -f $C/pgo/*.gcda
g++ -O3 -march=x86-64-v4 -std=c++17 -fPIC -Wall -Wno-unused-result -fprofile-generate -fprofile-update=atomic -c $C/lzma_parser.cpp -o $C/lzma_parser.v4.host.o
OBJS=$(cd $C && ls *.hip.o api_backend.o api_rzip.o driver.o shard.o shard_rccl.o stream_api.o unrzip.o lzma_parser.host.o md5.host.o lzma_dec.host.o stream_layer.host.o hashes.host.o filters.host.o api_hash.host.o stream_in.host.o lzma_parser.v4.host.o)
(cd $C && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../liblrzgpu.so $OBJS -lpthread -ldl $(gcc -print-file-name=libgcov.a))
python tools/pgo_train.py 2>&1 | tail -2
ls -la $C/*.gcda
cp $C/lzma_parser.v4.host.gcda "$ROOT/$C/pgo/lzma_parser.v4.host.gcda"
echo "profile recorded: $ROOT/$C/pgo/lzma_parser.v4.host.gcda"
