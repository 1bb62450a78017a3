"""The numerical argument of the two split-GEMM arms, on the CPU (scripts/study_split_gemm.py): with exact partial
products and f32 accumulation per matrix instruction, three bf16 parts / six products and two scaled fp16 parts / three
products are as close to float64 as f32 operands with f32 accumulation - over activations, tiny gradients, heavy tails
and operands whose elements are orders of magnitude apart.  (The device kernels are held to the same comparison against
the f32-MFMA kernels in tests/test_f16x2_gpu.py.)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))
import study_split_gemm as study  # noqa: E402


def test_split_products_are_as_close_to_float64_as_f32_accumulation():
    rows = study.run(M=96)
    assert len(rows) == 2 * len(study.CASES)
    for name, K, N, e32, e3, e2 in rows:
        assert e3 < max(2.0 * e32, 4e-7), (name, K, e32, e3)
        assert e2 < max(2.0 * e32, 4e-7), (name, K, e32, e2)
        assert e2 < 1e-6 and e3 < 1e-6


def test_power_of_two_scale_matches_the_device_rule():
    for m, want in [(1.0, 2.0 ** 14), (1.999, 2.0 ** 14), (2.0, 2.0 ** 13), (3e4, 1.0), (6e-8, 2.0 ** 38), (0.0, 2.0 ** 125),
                    (1e-45, 2.0 ** 125)]:
        s, inv = study.pow2_scale(np.array([m, -m / 3], dtype=np.float32))
        assert s == want, (m, s, want)
        assert s * inv == 1.0
        if m > 1e-30:
            assert 2.0 ** 14 <= m * s < 2.0 ** 15


def test_two_fp16_parts_carry_f32_precision_down_to_tiny_elements():
    rng = np.random.RandomState(1)
    x = (rng.randn(4096) * np.exp(rng.randn(4096) * 3)).astype(np.float32)
    s, inv = study.pow2_scale(x)
    h1, h2 = study.split_f16x2(x, s)
    back = (h1.astype(np.float64) + h2.astype(np.float64)) * inv
    top = np.abs(x).max()
    big = np.abs(x) > top * 2.0 ** -16
    assert np.abs(back[big] / x[big] - 1.0).max() < 2.0 ** -21          # elements within 2^16 of the maximum: 22+ bits
    assert np.abs(back[~big] - x[~big]).max() < top * 2.0 ** -36         # smaller elements: negligible against the maximum
