#!/bin/bash
# Builds the timing harness once per parser variant.  A variant is a directory under tools/_bin/variants/ holding
# its own lzma_parser.cpp (and optionally lzma_model.h / lzma_rangecoder.h; missing files are taken from csrc/);
# "current" = the working tree.  PROF=1 adds -DLZMA_PARSER_PROF (cycle laps); PGO=<file.gcda> builds the AVX-512
# object with that recorded profile.  Output: tools/_bin/pv_<variant>[_prof|_pgo].
set -e
cd "$(dirname "$0")/../.."
C=lrzip-next_amd/csrc
mkdir -p tools/_bin
for v in "$@"; do
  W=/tmp/pv_build_$v; rm -rf $W; mkdir -p $W/$C
  cp $C/*.h $W/$C/; cp $C/lzma_parser.cpp $W/$C/
  if [ "$v" != current ]; then cp tools/_bin/variants/$v/* $W/$C/; fi
  sfx=""; FL=""
  if [ -n "$PROF" ]; then FL="-DLZMA_PARSER_PROF"; sfx=_prof; fi
  if [ -n "$PGO" ]; then cp "$PGO" $W/$C/lzma_parser.v4.host.gcda; FL="$FL -fprofile-use -fprofile-correction -Wno-missing-profile -Wno-coverage-mismatch"; sfx=${sfx}_pgo; fi
  # the AVX-512 object exactly as csrc/Makefile builds it (path as given on the command line matters to -fprofile-use)
  (cd $W && g++ -O3 -g -march=x86-64-v4 -std=c++17 -fPIC -Wall -Wno-unused-result $FL -c $C/lzma_parser.cpp -o $C/lzma_parser.v4.host.o)
  (cd $W && g++ -O3 -march=x86-64-v3 -std=c++17 -fPIC -Wall -Wno-unused-result ${PROF:+-DLZMA_PARSER_PROF} -c $C/lzma_parser.cpp -o $C/lzma_parser.host.o)
  g++ -O2 -std=c++17 -no-pie -I$W/$C -o tools/_bin/pv_$v$sfx tools/parser_variants/harness.cpp $W/$C/lzma_parser.host.o $W/$C/lzma_parser.v4.host.o -lpthread
  echo "built tools/_bin/pv_$v$sfx"
done
