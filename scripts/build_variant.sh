#!/bin/bash
# usage: scripts/build_variant.sh NAME "-DKWS_GEMM_BK=32 -DKWS_GEMM_STAGGER=2"  -> variants/libkws_NAME.so (kernel A/B experiments)
set -e
cd "$(dirname "$0")/../speech_recognition_amd/csrc"
mkdir -p ../../variants build
make -s -j8 >/dev/null
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function $2 -c gemm.hip -o build/gemm_$1.o
objs=$(ls build/*.o | grep -v "gemm" | tr '\n' ' ')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../variants/libkws_$1.so $objs build/gemm_$1.o
echo built variants/libkws_$1.so
