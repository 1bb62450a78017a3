set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_fallback_paths_gpu.py tests/test_val_acc_gpu.py tests/test_net_gpu.py tests/test_bench_gpu.py -q > gpurun_out/r3b_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r3b_tests.log
tail -5 gpurun_out/r3b_tests.log
python bench.py > gpurun_out/r3b_bench.json 2> gpurun_out/r3b_bench.err; echo "bench rc=$?"
tail -3 gpurun_out/r3b_bench.err
bash scripts/pmc_stft.sh gpurun_out/r03_stft_sq.json > gpurun_out/r3b_pmc_stft.log 2>&1
tail -40 gpurun_out/r3b_pmc_stft.log
