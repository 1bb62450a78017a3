#!/bin/bash
# A/B of library builds under rocprofv3 --kernel-trace --stats:
#   scripts/ab_stats.sh <out.txt> "<python script + args>" <kernel name substring> [variants/libkws_X.so ...]
# The shipped library is always the first and the last arm; prints the average duration of every kernel whose name contains the substring.
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
out=$1; cmd=$2; pat=$3; shift 3
: > "$out"
arm() {
  local tag=$1 lib=$2
  rm -rf gpurun_out/_abst
  if [ -n "$lib" ]; then export KWS_LIB_PATH=$lib; else unset KWS_LIB_PATH; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/_abst -- python3 $cmd > gpurun_out/_abst.log 2>&1
  f=$(find gpurun_out/_abst -name "*kernel_stats.csv" | head -1)
  echo "-- $tag --" >> "$out"
  grep -o "ms_per_step': [0-9.]*" gpurun_out/_abst.log >> "$out"
  python3 scripts/kstat.py "$f" "$pat" >> "$out"
  rm -rf gpurun_out/_abst
}
arm "shipped library" ""
for v in "$@"; do arm "$v" "$v"; done
arm "shipped library (again)" ""
unset KWS_LIB_PATH
cat "$out"
