#!/bin/bash
# usage: scripts/build_variant.sh NAME "-DKWS_GEMM_BK=32 -DKWS_GEMM_STAGGER=2" [FILE]
#   -> variants/libkws_NAME.so with FILE.hip (default gemm) rebuilt under the given flags (kernel A/B experiments)
set -e
cd "$(dirname "$0")/../speech_recognition_amd/csrc"
f=${3:-gemm}
mkdir -p ../../variants build/variant
make -s -j8 >/dev/null
extra=""; [ "$f" = stft4 ] && extra="-fno-slp-vectorize"   # as the Makefile builds it
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -fvisibility=hidden --offload-arch=gfx950 -Wno-unused-function -ffp-contract=on $extra $2 -c $f.hip -o build/variant/${f}_$1.o
objs=$(ls build/*.o | grep -v "build/$f.o" | tr '\n' ' ')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../variants/libkws_$1.so $objs build/variant/${f}_$1.o
echo built variants/libkws_$1.so
