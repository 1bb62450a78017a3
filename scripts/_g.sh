cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py tests/test_processor_features_gpu.py tests/test_fallback_paths_gpu.py tests/test_logmfcc_gpu.py tests/test_fullsize_gpu.py -q -k "stft or feature or audio or c3_features or unaligned or mel" 2>&1 | tail -3
for rep in 1 2; do
echo "new:  $(python scripts/bench_stft.py 1024,80,60 2048,40,40 2>&1 | grep B= | tr '\n' '|')"
echo "prev: $(KWS_LIB_PATH=variants/libkws_prev.so python scripts/bench_stft.py 1024,80,60 2048,40,40 2>&1 | grep B= | tr '\n' '|')"
done
