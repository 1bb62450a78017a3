#!/bin/bash
# Profiling passes of one round on the GPU box (run through gpurun from the repo root):
#   scripts/profile_round.sh r03
# 1. rocprofv3 --kernel-trace --stats of the bench command        -> gpurun_out/<tag>_stats/
# 2. --pmc FETCH_SIZE and --pmc WRITE_SIZE in SEPARATE passes     -> gpurun_out/<tag>_pmc_traffic.json (scripts/pmc_traffic.py)
# 3. --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE (own pass)    -> gpurun_out/<tag>_pmc_mfma_busy.json
# 4. the same stats + traffic passes with KWS_GEMM_F16X2=1 (the fp16 x 2 A/B arm)      -> gpurun_out/<tag>_f16x2_*
#    (copy the arm's files to profiles/ as <tag>_f16x2_arm_kernel_stats_bench_b1024.csv and <tag>_f16x2_arm_pmc.json)
# 5. kernel stats of BASELINE configs[2] (C3) and configs[4] (C5): scripts/prof_c3.py / prof_c5.py -> <tag>_kernel_stats_c3.csv / _c5.csv
# PMC passes never share a run with trace domains other than --kernel-trace.  Copy the summaries into profiles/.
set -u
tag=${1:-r03}
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
BENCH="python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-val-acc --no-ab --no-configs"
mkdir -p gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_stats -- $BENCH > gpurun_out/${tag}_stats.log 2>&1
f=$(find gpurun_out/${tag}_stats -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" gpurun_out/${tag}_kernel_stats_bench_b1024.csv && python3 scripts/stats_sources.py gpurun_out/${tag}_kernel_stats_bench_b1024.csv
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/${tag}_pmc_$c -- python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-val-acc --no-ab --no-configs --profile-steps 0 > gpurun_out/${tag}_pmc_$c.log 2>&1
done
ff=$(find gpurun_out/${tag}_pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1)
fw=$(find gpurun_out/${tag}_pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1)
python3 scripts/pmc_traffic.py "$ff" "$fw" gpurun_out/${tag}_pmc_traffic.json > gpurun_out/${tag}_pmc_traffic.txt 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/${tag}_pmc_MFMA -- python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-val-acc --no-ab --no-configs --profile-steps 0 > gpurun_out/${tag}_pmc_MFMA.log 2>&1
fm=$(find gpurun_out/${tag}_pmc_MFMA -name "*counter_collection.csv" | head -1)
python3 scripts/pmc_mfma_busy.py "$fm" gpurun_out/${tag}_pmc_mfma_busy.json > gpurun_out/${tag}_pmc_mfma_busy.txt 2>&1
# 4. the fp16 x 2 arm (experiment 2, KWS_GEMM_F16X2=1): kernel stats and the two traffic passes of the same command
if [ -z "${KWS_PROFILE_NO_ARMS:-}" ]; then
  export KWS_GEMM_F16X2=1
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_f16x2_stats -- $BENCH > gpurun_out/${tag}_f16x2_stats.log 2>&1
  f=$(find gpurun_out/${tag}_f16x2_stats -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" gpurun_out/${tag}_f16x2_kernel_stats_bench_b1024.csv
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/${tag}_f16x2_pmc_$c -- python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-val-acc --no-ab --no-configs --profile-steps 0 > gpurun_out/${tag}_f16x2_pmc_$c.log 2>&1
  done
  ff=$(find gpurun_out/${tag}_f16x2_pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1)
  fw=$(find gpurun_out/${tag}_f16x2_pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1)
  python3 scripts/pmc_traffic.py "$ff" "$fw" gpurun_out/${tag}_f16x2_pmc_traffic.json > gpurun_out/${tag}_f16x2_pmc_traffic.txt 2>&1
  unset KWS_GEMM_F16X2
  rm -rf gpurun_out/${tag}_f16x2_stats gpurun_out/${tag}_f16x2_pmc_FETCH_SIZE gpurun_out/${tag}_f16x2_pmc_WRITE_SIZE
fi
# 5. C3 (32-class conv_1d_log_mfcc, batch 2048 training steps) and C5 (TTA x3 + plain inference, batch 4096)
for c in c3 c5; do
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_stats_$c -- python3 scripts/prof_$c.py > gpurun_out/${tag}_stats_$c.log 2>&1
  f=$(find gpurun_out/${tag}_stats_$c -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" gpurun_out/${tag}_kernel_stats_$c.csv && python3 scripts/stats_sources.py gpurun_out/${tag}_kernel_stats_$c.csv
  rm -rf gpurun_out/${tag}_stats_$c
done
# the raw traces are large: keep the summaries only
rm -rf gpurun_out/${tag}_stats gpurun_out/${tag}_pmc_FETCH_SIZE gpurun_out/${tag}_pmc_WRITE_SIZE gpurun_out/${tag}_pmc_MFMA
ls -la gpurun_out | grep ${tag}
