cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests -m gpu -q > gpurun_out/r3k_tests.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/r3k_tests.log
