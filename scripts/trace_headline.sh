#!/bin/bash
# One training step of the headline net (batch 1024, no generator beside it) as a launch-by-launch list:
#   scripts/trace_headline.sh <out.txt>      (rocprofv3 --kernel-trace of scripts/bench_conv1.py -> scripts/step_trace.py)
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
rm -rf gpurun_out/_trh
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/_trh -- python3 scripts/bench_conv1.py > gpurun_out/_trh.log 2>&1
t=$(find gpurun_out/_trh -name "*kernel_trace.csv" | head -1)
python3 scripts/step_trace.py "$t" "$1"
rm -rf gpurun_out/_trh
