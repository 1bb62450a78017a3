#!/bin/bash
# SQ counters of the STFT+mel kernel alone (scripts/bench_stft.py), one rocprofv3 --pmc pass per counter group.
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_INSTS_SMEM" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  tag=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d gpurun_out/stftpmc_$tag -- python3 scripts/bench_stft.py > gpurun_out/stftpmc_$tag.log 2>&1
  f=$(find gpurun_out/stftpmc_$tag -name "*counter_collection.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(sys.argv[1])):
    if "stft" in r["Kernel_Name"]:
        a = agg[(r["Kernel_Name"][:60], r["Counter_Name"])]
        a[0] += float(r["Counter_Value"]); a[1] += 1
for (k, c), (v, n) in sorted(agg.items()):
    print("%-50s %-28s avg/launch %.4g (n=%d)" % (k[-50:], c, v / n, n))
PY
  rm -rf gpurun_out/stftpmc_$tag
done
