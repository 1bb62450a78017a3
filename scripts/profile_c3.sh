#!/bin/bash
# Profiling passes of BASELINE configs[2] (C3: 32-class conv_1d_log_mfcc, batch 2048) on the GPU box, DEFAULT schedule only:
#   scripts/profile_c3.sh r06
# 1. rocprofv3 --kernel-trace --stats of scripts/prof_c3.py (35 steps incl. 5 warm-up)  -> gpurun_out/<tag>_kernel_stats_c3.csv
# 2. --pmc FETCH_SIZE and --pmc WRITE_SIZE in SEPARATE passes (11 steps)               -> gpurun_out/<tag>_pmc_traffic_c3.json
# 3. --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE (own pass)                         -> gpurun_out/<tag>_pmc_mfma_busy_c3.json
# PMC passes never share a run with trace domains other than --kernel-trace.  Copy the summaries into profiles/.
set -u
tag=${1:-r06}
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_stats_c3 -- python3 scripts/prof_c3.py 30 > gpurun_out/${tag}_stats_c3.log 2>&1
f=$(find gpurun_out/${tag}_stats_c3 -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" gpurun_out/${tag}_kernel_stats_c3.csv && python3 scripts/stats_sources.py gpurun_out/${tag}_kernel_stats_c3.csv
t=$(find gpurun_out/${tag}_stats_c3 -name "*kernel_trace.csv" | head -1)
[ -n "$t" ] && python3 scripts/step_trace.py "$t" gpurun_out/${tag}_step_trace_c3.txt
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/${tag}_c3_pmc_$c -- python3 scripts/prof_c3.py 6 > gpurun_out/${tag}_c3_pmc_$c.log 2>&1
done
ff=$(find gpurun_out/${tag}_c3_pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1)
fw=$(find gpurun_out/${tag}_c3_pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1)
python3 scripts/pmc_traffic.py "$ff" "$fw" gpurun_out/${tag}_pmc_traffic_c3.json "scripts/prof_c3.py 6 (C3, batch 2048, 11 steps incl. warm-up)" > gpurun_out/${tag}_pmc_traffic_c3.txt 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/${tag}_c3_pmc_MFMA -- python3 scripts/prof_c3.py 6 > gpurun_out/${tag}_c3_pmc_MFMA.log 2>&1
fm=$(find gpurun_out/${tag}_c3_pmc_MFMA -name "*counter_collection.csv" | head -1)
python3 scripts/pmc_mfma_busy.py "$fm" gpurun_out/${tag}_pmc_mfma_busy_c3.json "scripts/prof_c3.py 6 (C3, batch 2048)" > gpurun_out/${tag}_pmc_mfma_busy_c3.txt 2>&1
rm -rf gpurun_out/${tag}_stats_c3 gpurun_out/${tag}_c3_pmc_FETCH_SIZE gpurun_out/${tag}_c3_pmc_WRITE_SIZE gpurun_out/${tag}_c3_pmc_MFMA
ls -la gpurun_out | grep ${tag}
