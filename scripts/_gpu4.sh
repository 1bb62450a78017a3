set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
python -m pytest tests/test_kernels_gpu.py -q -k "gemm or tn or wgrad" > gpurun_out/r3d_tests.log 2>&1; tail -3 gpurun_out/r3d_tests.log
python scripts/bench_gemm.py > gpurun_out/r3d_gemm_per_layer.txt 2>&1
tail -13 gpurun_out/r3d_gemm_per_layer.txt
python bench.py --no-cpu-baseline --no-val-acc --no-configs > gpurun_out/r3d_bench.json 2> gpurun_out/r3d_bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3d_bench.json').read())
print(d['value'], d['ms_per_step'])
print('roof', d['roofline']['avg_launch_us'], d['roofline']['frac'])
for e in d['roofline_stages']: print(e['family'], round(e['avg_launch_us'],1), round(e['frac'],3))
PY
