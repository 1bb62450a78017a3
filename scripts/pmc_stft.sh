#!/bin/bash
# SQ counters of the STFT+mel kernel alone (scripts/bench_stft.py at batch 1024, 80 mel bands, 60 coefficients), one
# rocprofv3 --pmc pass per counter group (8 SQ slots per pass; counters never share a run with a trace domain other than
# --kernel-trace).  usage: scripts/pmc_stft.sh [out.json]   -> gpurun_out/r03_stft_sq.json by default
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
out=${1:-gpurun_out/r03_stft_sq.json}
mkdir -p gpurun_out
rm -f gpurun_out/stftpmc_all.csv
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_INSTS_SMEM" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VMEM_WR SQ_WAVES SQ_LDS_ADDR_CONFLICT"; do
  tag=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d gpurun_out/stftpmc_$tag -- python3 scripts/bench_stft.py 1024,80,60 > gpurun_out/stftpmc_$tag.log 2>&1
  f=$(find gpurun_out/stftpmc_$tag -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then
    if [ ! -f gpurun_out/stftpmc_all.csv ]; then head -1 "$f" > gpurun_out/stftpmc_all.csv; fi
    grep stft4_kernel "$f" >> gpurun_out/stftpmc_all.csv
  else
    echo "group '$grp' produced no counter file (an unknown counter name?):"; tail -5 gpurun_out/stftpmc_$tag.log
  fi
  rm -rf gpurun_out/stftpmc_$tag
done
python3 scripts/pmc_stft.py gpurun_out/stftpmc_all.csv "$out"
