"""The oracle's floating-point restatements against INDEPENDENT third-party implementations of the same functions that
ship in this image (TensorFlow itself does not): a second line of pinning besides the reference-generated fixtures of
tests/golden/ (which cover control logic, constants and tools, not tensors).

  * tf.contrib.signal.linear_to_mel_weight_matrix (reference input_data.py:369-373) <-> transformers.audio_utils
    .mel_filter_bank(mel_scale="htk", triangularize_in_mel_space=True, norm=None) - the Hugging Face port of that TF op;
  * tf.contrib.signal.hann_window(periodic=True) / stft (input_data.py:361-366) <-> scipy.signal.get_window,
    torch.hann_window, torch.stft;
  * tf.spectral.dct type II inside mfccs_from_log_mel_spectrograms (input_data.py:376-381) <-> scipy.fft.dct;
  * Keras 2.1.2 RMSprop / SGD(momentum) update rules (train.py:49-52) <-> torch.optim.RMSprop / SGD;
  * Keras BatchNormalization training forward, categorical cross-entropy on softmax outputs, sklearn-style log_loss
    (callbacks.py:6-10) <-> torch.nn.functional / sklearn.metrics.
"""
import numpy as np
import pytest
import torch

from oracle import features as OF
from oracle import layers as OL


@pytest.mark.parametrize("n_mel", [80, 40])
def test_mel_matrix_matches_the_hugging_face_port_of_the_tf_op(n_mel):
    audio_utils = pytest.importorskip("transformers.audio_utils")
    theirs = audio_utils.mel_filter_bank(num_frequency_bins=257, num_mel_filters=n_mel, min_frequency=80.0,
                                         max_frequency=7600.0, sampling_rate=16000, norm=None, mel_scale="htk",
                                         triangularize_in_mel_space=True)
    ours = OF.linear_to_mel_weight_matrix(n_mel, 257, 16000, 80.0, 7600.0)
    assert ours.shape == theirs.shape == (257, n_mel)
    # ours runs in float32 like the TF graph, theirs in float64: a few float32 ulps of the band edges
    np.testing.assert_allclose(ours, theirs, atol=3e-5)
    assert np.array_equal(ours > 1e-4, theirs > 1e-4) or np.abs(ours - theirs)[(ours > 1e-4) != (theirs > 1e-4)].max() < 3e-5
    assert np.all(ours[0] == 0.0)                    # the DC bin is dropped and padded back as a zero row


@pytest.mark.parametrize("n", [480, 400, 240, 481])
def test_periodic_hann_window(n):
    scipy_signal = pytest.importorskip("scipy.signal")
    ours = OF.hann_periodic(n)
    if n % 2 == 0:                                   # for even n TF's periodic window is the textbook one
        np.testing.assert_allclose(ours, scipy_signal.get_window("hann", n, fftbins=True), atol=5e-7)
        np.testing.assert_allclose(ours, torch.hann_window(n, periodic=True).numpy(), atol=5e-7)
    else:                                            # odd n: TF 1.4 divides by n - 1 (window_ops._raised_cosine_window)
        np.testing.assert_allclose(ours, scipy_signal.get_window("hann", n, fftbins=False), atol=5e-7)


@pytest.mark.parametrize("window,step", [(480, 160), (400, 160), (240, 80)])
def test_stft_magnitude_matches_torch_stft(window, step):
    rng = np.random.RandomState(window)
    x = rng.randn(3, 16000) * 0.1
    tables = OF.tables_path_b(window_size=window)
    n_fft = tables["fft_length"]
    ours = OF.stft_magnitude(x, tables, frame_step=step)
    # TF zero-pads a windowed frame at its END, torch centres the short window inside n_fft: the same frame shifted by
    # (n_fft - window) / 2 samples, which leaves the magnitude alone
    pad = (n_fft - window) // 2
    xt = torch.nn.functional.pad(torch.from_numpy(x), (pad, n_fft - window - pad))
    theirs = torch.stft(xt, n_fft=n_fft, hop_length=step, win_length=window,
                        window=torch.hann_window(window, periodic=True, dtype=torch.float64), center=False,
                        return_complex=True).abs().transpose(1, 2).numpy()
    assert ours.shape == theirs.shape == (3, 1 + (16000 - window) // step, n_fft // 2 + 1)
    np.testing.assert_allclose(ours, theirs, atol=2e-5 * np.abs(theirs).max())


@pytest.mark.parametrize("n_mel,keep", [(80, 60), (40, 40), (40, 13)])
def test_dct_matrix_matches_scipy(n_mel, keep):
    scipy_fft = pytest.importorskip("scipy.fft")
    x = np.random.RandomState(n_mel + keep).randn(7, n_mel)
    ours = x @ OF.dct2_matrix(n_mel, keep)
    unnormalised = scipy_fft.dct(x, type=2, norm=None, axis=-1)[:, :keep]
    np.testing.assert_allclose(ours, unnormalised / np.sqrt(2.0 * n_mel), atol=1e-12)
    # ... which is the orthonormal DCT-II except for coefficient 0 (TF 1.4 scales every coefficient by rsqrt(2 M))
    ortho = scipy_fft.dct(x, type=2, norm="ortho", axis=-1)[:, :keep]
    np.testing.assert_allclose(ours[:, 1:], ortho[:, 1:], atol=1e-12)
    np.testing.assert_allclose(ours[:, 0], ortho[:, 0] * np.sqrt(2.0), atol=1e-12)


def test_whole_feature_path_against_the_third_party_pieces():
    """log(|stft| . mel + 1e-6) . dct assembled from torch.stft, the Hugging Face mel matrix and scipy's DCT"""
    audio_utils = pytest.importorskip("transformers.audio_utils")
    scipy_fft = pytest.importorskip("scipy.fft")
    x = np.random.RandomState(5).randn(2, 16000) * 0.05
    ours = OF.features(x, OF.tables_path_b(480, 80, 60))
    xt = torch.nn.functional.pad(torch.from_numpy(x), (16, 16))
    mag = torch.stft(xt, n_fft=512, hop_length=160, win_length=480,
                     window=torch.hann_window(480, periodic=True, dtype=torch.float64), center=False,
                     return_complex=True).abs().transpose(1, 2).numpy()
    mel = audio_utils.mel_filter_bank(257, 80, 80.0, 7600.0, 16000, norm=None, mel_scale="htk",
                                      triangularize_in_mel_space=True)
    theirs = scipy_fft.dct(np.log(mag @ mel + 1e-6), type=2, norm=None, axis=-1)[..., :60] / np.sqrt(160.0)
    assert ours.shape == theirs.shape == (2, 98, 60)
    np.testing.assert_allclose(ours, theirs, atol=2e-3)      # the device parity bar for this tensor is 2e-3 as well


def test_rmsprop_rule_matches_torch():
    rng = np.random.RandomState(0)
    p0 = rng.randn(257).astype(np.float64)
    tp = torch.nn.Parameter(torch.from_numpy(p0.copy()))
    opt = torch.optim.RMSprop([tp], lr=1e-3, alpha=0.9, eps=1e-8, weight_decay=0.0, momentum=0.0, centered=False)
    p, a = p0.copy(), np.zeros_like(p0)
    for step in range(5):
        g = rng.randn(257) * 10.0 ** (step - 2)
        tp.grad = torch.from_numpy(g.copy())
        opt.step()
        p, a = OL.rmsprop_step(p, g, a, 1e-3, rho=0.9, eps=1e-8)
        np.testing.assert_allclose(p, tp.detach().numpy(), rtol=1e-12, atol=1e-14)


def test_sgd_momentum_rule_matches_torch():
    """Keras: v = m v - lr g, p += v.  torch: b = m b + g, p -= lr b.  The same trajectory while lr is constant."""
    rng = np.random.RandomState(1)
    p0 = rng.randn(100).astype(np.float64)
    tp = torch.nn.Parameter(torch.from_numpy(p0.copy()))
    opt = torch.optim.SGD([tp], lr=0.01, momentum=0.9, nesterov=False)
    p, v = p0.copy(), np.zeros_like(p0)
    for _ in range(6):
        g = rng.randn(100)
        tp.grad = torch.from_numpy(g.copy())
        opt.step()
        p, v = OL.sgd_momentum_step(p, g, v, 0.01, momentum=0.9)
        np.testing.assert_allclose(p, tp.detach().numpy(), rtol=1e-12, atol=1e-14)


def test_batchnorm_training_forward_matches_torch():
    rng = np.random.RandomState(2)
    y = rng.randn(6, 50, 24) * 3.0 + 1.0             # [batch, time, channels]: statistics over batch x time
    gamma, beta = rng.rand(24) + 0.5, rng.randn(24)
    out = OL.bn_train_fwd(y, gamma, beta, eps=OL.BN_EPS)
    ours = out[0] if isinstance(out, tuple) else out
    theirs = torch.nn.functional.batch_norm(torch.from_numpy(y).permute(0, 2, 1), None, None, torch.from_numpy(gamma),
                                            torch.from_numpy(beta), training=True, eps=OL.BN_EPS).permute(0, 2, 1).numpy()
    np.testing.assert_allclose(ours, theirs, atol=1e-10)


def test_log_loss_matches_sklearn():
    metrics = pytest.importorskip("sklearn.metrics")
    rng = np.random.RandomState(3)
    logits = rng.randn(64, 12) * 3.0
    probs = OL.softmax(logits)
    labels = rng.randint(0, 12, size=64)
    onehot = np.eye(12)[labels]
    np.testing.assert_allclose(OL.log_loss(onehot, probs), metrics.log_loss(labels, probs, labels=list(range(12))), rtol=1e-12)
    np.testing.assert_allclose(probs, torch.softmax(torch.from_numpy(logits), dim=1).numpy(), atol=1e-14)


def test_label_smoothed_cross_entropy_matches_torch():
    rng = np.random.RandomState(4)
    logits = rng.randn(32, 12)
    labels = rng.randint(0, 12, size=32)
    onehot = np.eye(12)[labels]
    mean, per, _ = OL.smooth_cce_fwd_bwd(OL.softmax(logits), onehot, label_smoothing=0.1)
    theirs = torch.nn.functional.cross_entropy(torch.from_numpy(logits), torch.from_numpy(labels), label_smoothing=0.1,
                                               reduction="none").numpy()
    np.testing.assert_allclose(per, theirs, atol=1e-6)       # Keras clips p to [1e-7, 1 - 1e-7] before the log
    np.testing.assert_allclose(mean, theirs.mean(), atol=1e-6)


def test_path_a_filterbank_matches_the_hugging_face_matrix_but_for_tf_kernels_start_index():
    """audio.py's path A (TF 1.4 C++ MfccMelFilterbank, 40 channels, 20 - 4000 Hz): triangles in HTK-mel space like the
    path-B op, so the same third-party matrix applies - except that the TF kernel skips the spectrum bins below
    start_index = int(1.5 + lower / hz_per_bin) (= 2 here: bin 1 at 31.25 Hz lies inside the first triangle but is dropped)."""
    audio_utils = pytest.importorskip("transformers.audio_utils")
    theirs = audio_utils.mel_filter_bank(num_frequency_bins=257, num_mel_filters=40, min_frequency=20.0, max_frequency=4000.0,
                                         sampling_rate=16000, norm=None, mel_scale="htk", triangularize_in_mel_space=True)
    ours = OF.mfcc_mel_filterbank_dense(257, 16000.0, 40, 20.0, 4000.0)
    diff = np.abs(ours - theirs)
    rows = sorted(set(np.argwhere(diff > 1e-9)[:, 0].tolist()))
    assert rows == [1]                               # the one bin the TF kernel's start index excludes
    assert np.all(ours[:2] == 0.0) and theirs[1, 0] > 0.3
    np.testing.assert_allclose(np.delete(ours, 1, axis=0), np.delete(theirs, 1, axis=0), atol=1e-9)


def test_path_a_dct_matches_scipy():
    """TF 1.4 MfccDct: sqrt(2/N) cos(pi i (j + 0.5) / N) = the orthonormal DCT-II except for coefficient 0 (no 1/sqrt 2)"""
    scipy_fft = pytest.importorskip("scipy.fft")
    x = np.random.RandomState(11).randn(6, 40)
    for keep in (40, 13):
        ours = x @ OF.mfcc_dct_matrix(40, keep)
        ortho = scipy_fft.dct(x, type=2, norm="ortho", axis=-1)[:, :keep]
        np.testing.assert_allclose(ours[:, 1:], ortho[:, 1:], atol=1e-12)
        np.testing.assert_allclose(ours[:, 0], ortho[:, 0] * np.sqrt(2.0), atol=1e-12)


@pytest.mark.parametrize("path", ["B", "A"])
def test_batched_f32_feature_path_of_the_cpu_baseline_matches_the_float64_oracle(path):
    """oracle.features.features_batched (bench.py's CPU baseline B2: scipy.fft.rfft(workers) + two GEMMs, float32) computes
    what features() computes in float64 - the baseline times the right arithmetic."""
    rng = np.random.RandomState(11)
    t = np.arange(16000) / 16000.0
    x = (rng.randn(6, 16000) * 0.0774 + 0.05 * np.sin(2 * np.pi * 600 * t)[None]).astype(np.float32)
    tables = OF.tables_path_b(480, 80, 60) if path == "B" else OF.tables_path_a(480, 16000, 40, 40)
    ref = OF.features(x.astype(np.float64), tables, 160)
    for workers in (1, 2):
        got = OF.features_batched(x, tables, 160, workers=workers)
        assert got.shape == ref.shape and got.dtype == np.float32
        assert np.abs(got - ref).max() < 2e-3          # the device kernels' own bar (tests/test_kernels_gpu.py)
