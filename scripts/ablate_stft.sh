#!/bin/bash
# Ablation timing of stft4_kernel (timing only - the ablated kernels compute wrong results): build the variants with
#   for b in 1 2 4 8 16 32 64 127; do scripts/build_variant.sh abl$b "-DKWS_STFT_ABL=$b" stft4; done
# and run this on the GPU box.  Prints the batch-1024 line of scripts/bench_stft.py per variant.
cd "$(dirname "$0")/.."
echo "full:     $(python scripts/bench_stft.py 2>/dev/null | grep 'B=1024')"
for b in 1 2 4 8 16 32 64 127; do
  echo "without $b: $(KWS_LIB_PATH=variants/libkws_abl$b.so python scripts/bench_stft.py 2>/dev/null | grep 'B=1024')"
done
