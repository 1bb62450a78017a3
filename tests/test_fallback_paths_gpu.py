"""The library's shape-selected fallback paths, each reached the way a caller reaches it (no environment switches):

  * filter_mult = 2 (model.py:775: `filter_mult` argument): 256 first-convolution channels -> the generic gathered GEMMs of
    gemm.hip instead of conv1.hip, 1024-wide late layers;
  * a clip length other than 16000 (prepare_model_settings with another clip_duration_ms): T != 9 at the tail -> the generic
    tail kernel, other row counts everywhere;
  * an STFT input that is not 16-byte aligned -> stft4's 8-byte PCM loads;
  * a mel shape stft4 does not instantiate -> the generic STFT kernel (the other output kinds run it too,
    tests/test_kernels_gpu.py).
Each against the float64 oracle."""
import ctypes

import numpy as np
import pytest
import torch

from oracle import features as OF
from speech_recognition_amd import _lib

from test_kernels_gpu import _plan
from test_net_gpu import _batch, _check_grads, _pair

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("kw,B", [(dict(filter_mult=2), 3), (dict(input_size=12000), 5), (dict(input_size=20000), 2)])
def test_train_step_on_fallback_shapes_matches_oracle(kw, B):
    ora, net = _pair(**kw)
    L = kw.get("input_size", 16000)
    x, y = _batch(B, 12, 7 + B, L=L)
    pr = net.predict(torch.from_numpy(x).cuda()).cpu().numpy()          # the inference program on the same shapes (before the
    ref = ora.forward(x.astype(np.float64), training=False)              # training step moves the BN moving statistics)
    assert np.abs(pr - ref).max() < 1e-5 and np.array_equal(pr.argmax(1), ref.argmax(1))
    probs = net.train_fwd_bwd(torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda(), seed=99, step=2)
    torch.cuda.synchronize()
    loss, p, grads, cache = _check_grads(ora, net, x, y, 99, 2, B, tol=1e-4)
    got = probs.cpu().numpy()
    assert np.abs(got - p).max() < 2e-5
    assert np.array_equal(got.argmax(1), p.argmax(1))
    assert abs(float(net.metrics.cpu().numpy()[0]) / B - loss) < 2e-5


@pytest.mark.parametrize("nc", [30, 32])
def test_train_step_with_other_class_counts_matches_oracle(nc):
    """30 classes (the 30 words of the data set; rows of the classifier kernel are not whole 16-byte vectors) and 32 (more than
    the 16 the vectorised classifier passes hold in registers): the tail kernel's generic classifier loops, and its one-wave
    softmax / loss reductions with more lanes in use"""
    B = 4
    ora, net = _pair(num_classes=nc)
    x, y = _batch(B, nc, 11 + nc)
    probs = net.train_fwd_bwd(torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda(), seed=7, step=1)
    torch.cuda.synchronize()
    loss, p, grads, cache = _check_grads(ora, net, x, y, 7, 1, B, tol=1e-4)
    got = probs.cpu().numpy()
    assert np.abs(got - p).max() < 2e-5 and np.array_equal(got.argmax(1), p.argmax(1))
    assert abs(float(net.metrics.cpu().numpy()[0]) / B - loss) < 2e-5
    assert float(net.metrics.cpu().numpy()[1]) == float((p.argmax(1) == y.argmax(1)).sum())


def _features(plan, dx, B, L, width):
    lib = _lib.load()
    F = lib.kws_stft_num_frames(plan, L)
    out = torch.full((B, F, width), float("nan"), device="cuda")
    _lib.call("kws_stft_mel_f32", plan, _lib.ptr(dx), B, L, _lib.ptr(out), 0, _lib.stream_ptr())
    torch.cuda.synchronize()
    return out.cpu().numpy()


def test_stft_features_from_an_unaligned_batch():
    """clips that start 8 bytes into a 16-byte line, and a clip length with L % 4 = 2: the 8-byte-load form of stft4"""
    rng = np.random.RandomState(4)
    t = OF.tables_path_b(480, 80, 60)
    plan = _plan(t, 160, 80, 60)
    for L, off in ((16000, 2), (16002, 0)):
        B = 5
        x = (rng.randn(B, L) * 0.0774).astype(np.float32)
        buf = torch.zeros(B * L + 8, device="cuda")
        dx = buf[off:off + B * L].view(B, L)
        dx.copy_(torch.from_numpy(x))
        assert dx.data_ptr() % 16 == (8 if off == 2 else 0)
        got = _features(plan, dx, B, L, 60)
        ref = OF.features(x.astype(np.float64), t, 160)
        assert got.shape == ref.shape and np.all(np.isfinite(got))
        assert np.abs(got - ref).max() < 2e-3      # the bar of tests/test_kernels_gpu.py::test_stft_mel_features
    _lib.load().kws_stft_plan_destroy(plan)


def test_stft_features_of_a_mel_shape_stft4_does_not_instantiate():
    """64 mel bands, 32 coefficients: four lane groups of bands - no stft4 instance -> the generic kernel"""
    rng = np.random.RandomState(5)
    t = OF.tables_path_b(480, 64, 32)
    plan = _plan(t, 160, 64, 32)
    B, L = 4, 16000
    x = (rng.randn(B, L) * 0.0774).astype(np.float32)
    got = _features(plan, torch.from_numpy(x).cuda(), B, L, 32)
    ref = OF.features(x.astype(np.float64), t, 160)
    assert np.abs(got - ref).max() < 2e-3      # the bar of tests/test_kernels_gpu.py::test_stft_mel_features
    _lib.load().kws_stft_plan_destroy(plan)
