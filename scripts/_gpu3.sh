set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
python scripts/bench_gemm.py > gpurun_out/r3c_gemm_per_layer.txt 2>&1
tail -13 gpurun_out/r3c_gemm_per_layer.txt
bash scripts/build_variant.sh wsstamp "-DKWS_GEMM_STAMP" gemm > gpurun_out/r3c_build.log 2>&1; tail -2 gpurun_out/r3c_build.log
for s in "406528 128 128" "203776 128 192" "201728 192 192" "101376 192 256" "99328 256 256" "50176 256 320" "48128 320 320" "24576 320 384" "22528 384 384" "11264 384 512" "9216 512 512"; do
  KWS_LIB_PATH=variants/libkws_wsstamp.so python scripts/stamps_tn.py $s 2>&1 | tail -1
done > gpurun_out/r3c_tn_stamps.txt
cat gpurun_out/r3c_tn_stamps.txt
# kernel-only times of the TN launches (reduce separate)
cat > /tmp/tn_only.py <<'PY'
import sys, os
sys.path.insert(0, os.getcwd())
import torch
from speech_recognition_amd import _lib
lib = _lib.load(); S = _lib.stream_ptr()
shapes = [(397,128,128),(199,128,192),(197,192,192),(99,192,256),(97,256,256),(49,256,320),(47,320,320),(24,320,384),(22,384,384),(11,384,512),(9,512,512)]
for L, K, N in shapes:
    M = 1024 * L
    A = torch.randn(M, K, device='cuda'); G = torch.randn(M, N, device='cuda'); dW = torch.empty(K, N, device='cuda')
    ws = torch.empty(int(lib.kws_gemm_tn_workspace_floats(M, K, N)), device='cuda')
    for _ in range(6):
        _lib.call("kws_gemm_tn_f32", _lib.ptr(A), _lib.ptr(G), _lib.ptr(dW), M, K, N, _lib.ptr(ws), S)
    torch.cuda.synchronize()
PY
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r3c_tn_trace -- python3 /tmp/tn_only.py > gpurun_out/r3c_tn_trace.log 2>&1
f=$(find gpurun_out/r3c_tn_trace -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY' > gpurun_out/r3c_tn_kernel_times.txt
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if 'gemm_tn' in r['Kernel_Name'] or 'reduce_slabs' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
i = 0
out = []
for r in rows:
    out.append((r['Kernel_Name'][:60], (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, r.get('Grid_Size'), r.get('LDS_Block_Size'), r.get('VGPR_Count')))
# 6 launches per shape (tn + reduce(s)): print the last of each group
for o in out: print("%-62s %8.1f us grid %s lds %s vgpr %s" % o)
PY
grep -n "gemm_tn" gpurun_out/r3c_tn_kernel_times.txt | awk 'NR%6==0' 
rm -rf gpurun_out/r3c_tn_trace
