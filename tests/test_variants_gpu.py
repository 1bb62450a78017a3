"""The one A/B arm the library keeps: the pointwise GEMMs as scaled fp16 x 2 split products.  The arm
is a property of a net handle (kws_net_set_gemm_mode); KWS_GEMM_F16X2=1 makes it the default of every DeviceNet of a
process, so the whole-network parity suites - incl. the batch-1024 step against the float64 oracle - are re-run in a
child process with it selected.  (Round 2 kept eight such variants; the losers were removed in round 3.  The paths that
are real fallbacks - generic first convolution, generic tail, 8-byte STFT loads, 4-wave GEMMs for ragged shapes - are
selected by shape and covered by direct tests: tests/test_fallback_paths_gpu.py.)"""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


def test_fp16x2_arm_passes_the_network_parity_suites(repo_root):
    e = dict(os.environ, KWS_GEMM_F16X2="1")
    files = ["tests/test_net_gpu.py", "tests/test_fullsize_gpu.py", "-k", "not stft and not c3 and not augment"]
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider"] + files, cwd=repo_root,
                       env=e, capture_output=True, text=True, timeout=900)
    tail = "\n".join(r.stdout.splitlines()[-15:])
    assert r.returncode == 0, tail
    assert " passed" in tail and "failed" not in tail
