"""The A/B paths the library keeps behind environment variables stay correct: the kernel and
network parity suites are re-run in a child process with every alternative selected (the switches are read once
per process, so they cannot be flipped inside this one)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("env", [
    {"KWS_GEMM_PERSIST": "1", "KWS_GEMM_TN_V1": "1", "KWS_TAIL_GENERIC": "1", "KWS_CONV1_GENERIC": "1"},
    {"KWS_OVERLAP": "1"},
    {"KWS_STFT_V2": "1"},
    {"KWS_STFT_V3": "1"},
    {"KWS_STFT_F32PASS": "1", "KWS_GEMM_NO_HALF": "1"},   # stft4 with the f32 first pass; NN GEMM tails in whole tiles
    {"KWS_STFT_LD8": "1"},        # stft4 with 8-byte PCM loads (the row dealing of unaligned frames)
    {"KWS_GEMM_BF16X3": "1"},     # experiment: the pointwise GEMMs as bf16 x 3 split products
    {"KWS_GEMM_F16X2": "1"},      # experiment 2: the pointwise GEMMs as scaled fp16 x 2 split products
])
def test_alternative_paths_pass_the_parity_suites(repo_root, env):
    e = dict(os.environ)
    e.update(env)
    stft_variant = any(k.startswith("KWS_STFT_") for k in env)
    files = ["tests/test_kernels_gpu.py", "tests/test_net_gpu.py"] if not stft_variant else \
        ["tests/test_kernels_gpu.py", "tests/test_logmfcc_gpu.py", "tests/test_fullsize_gpu.py",
         "tests/test_processor_features_gpu.py", "-k", "stft or c3 or audio or gemm_nn"]
    if "KWS_GEMM_BF16X3" in env or "KWS_GEMM_F16X2" in env:      # the whole-network parity suites incl. the batch-1024 step against the float64 oracle
        files = ["tests/test_net_gpu.py", "tests/test_fullsize_gpu.py", "-k", "not stft and not c3 and not augment"]
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider"] + files, cwd=repo_root,
                       env=e, capture_output=True, text=True, timeout=900)
    tail = "\n".join(r.stdout.splitlines()[-15:])
    assert r.returncode == 0, tail
    assert " passed" in tail and "failed" not in tail
