#!/bin/bash
# First-convolution kernels, A/B of library builds under rocprofv3 (the measurement of profiles/r0N_conv1_*.txt):
#   scripts/ab_conv1.sh <out.txt> [variants/libkws_X.so ...]      (the shipped library is always the first and the last arm)
# 24 training steps of the raw-waveform net at batch 1024 per arm (scripts/bench_conv1.py); prints every conv1 kernel's average.
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
out=$1; shift
: > "$out"
arm() {
  local tag=$1 lib=$2
  rm -rf gpurun_out/_c1ab
  if [ -n "$lib" ]; then export KWS_LIB_PATH=$lib; else unset KWS_LIB_PATH; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/_c1ab -- python3 scripts/bench_conv1.py > gpurun_out/_c1ab.log 2>&1
  f=$(find gpurun_out/_c1ab -name "*kernel_stats.csv" | head -1)
  echo "-- $tag --" >> "$out"
  python3 scripts/kstat.py "$f" conv1 >> "$out"
  rm -rf gpurun_out/_c1ab
}
arm "shipped library" ""
for v in "$@"; do arm "$v" "$v"; done
arm "shipped library (again)" ""
unset KWS_LIB_PATH
cat "$out"
