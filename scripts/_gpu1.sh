set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests -m gpu -q > gpurun_out/r3a_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r3a_tests.log
tail -5 gpurun_out/r3a_tests.log
