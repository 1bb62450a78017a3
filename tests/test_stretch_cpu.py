"""CPU checks of oracle/stretch.py (librosa 0.5.x time_stretch restated; parity unpinned - librosa is not
installed here).  Independent cross-checks: scipy.signal's STFT for the analysis stage, perfect
reconstruction for the synthesis stage, and the closed-form properties of the vocoder loop."""
import numpy as np
import scipy.signal

from oracle import stretch as OS


def clip(seed, L=16000):
    rng = np.random.RandomState(seed)
    return (rng.randn(L) * 0.0774).clip(-1, 1)


def test_stft_matches_scipy():
    y = clip(0)
    D = OS.stft(y)
    assert D.shape == (1025, 32)                  # 1 + 16000 // 512 centred frames
    yp = np.pad(y, 1024, mode='reflect')
    _, _, Z = scipy.signal.stft(yp, window='hann', nperseg=2048, noverlap=1536, boundary=None, padded=False)
    assert np.abs(Z * OS.hann_periodic().sum() - D).max() < 1e-12
    assert np.allclose(OS.hann_periodic(), scipy.signal.get_window('hann', 2048, fftbins=True), atol=1e-15)


def test_istft_inverts_stft():
    y = clip(1)
    r = OS.istft(OS.stft(y))
    assert len(r) == 512 * 31 and np.abs(r - y[:len(r)]).max() < 1e-12


def test_time_steps_and_lengths():
    idx, alpha = OS.time_steps(32, 0.9)
    assert len(idx) == 36 and idx[-1] == 31 and abs(alpha[-1] - 0.5) < 1e-12
    assert np.all(np.diff(idx) >= 0) and np.all(np.diff(idx) <= 1)
    assert OS.stretched_length(16000, 0.9) == 17920 == len(OS.time_stretch(clip(2), 0.9))
    assert len(OS.time_stretch(clip(2), 0.9, length_mode='round')) == 17778     # librosa >= 0.7 variant
    assert OS.stretched_length(16000, 1.0) == 15872


def test_rate_one_is_identity_and_phase_advance_is_irrelevant():
    y = clip(3)
    r = OS.time_stretch(y, 1.0)
    assert np.abs(r - y[:len(r)]).max() < 1e-10
    # exp(1j * acc) only sees angle(c1) - angle(c0) modulo 2 pi: the conjugated STFT of librosa 0.5 (the
    # "DPWE" convention) gives the same signal
    D = OS.stft(y)
    a = OS.istft(OS.phase_vocoder(D, 0.9))
    b = OS.istft(np.conj(OS.phase_vocoder(np.conj(D), 0.9)))
    assert np.abs(a - b).max() < 1e-9


def test_reference_rounding_noise_is_large():
    """librosa's float32 accumulator reaches ~6e4 rad: its own result is only good to ~5e-4 (16 int16 steps),
    which bounds what 'parity' with the reference can mean for this row."""
    y = clip(4)
    exact = OS.time_stretch(y, 0.9)
    lit = OS.time_stretch(y, 0.9, literal_f32=True)
    e = np.abs(lit - exact).max()
    assert 2e-5 < e < 5e-3


def test_wav_round_trip_semantics():
    y = clip(5)
    pcm = np.int16(y * 32767)
    out = OS.tta_slow_clip(pcm)
    assert out.dtype == np.float32 and out.shape == (16000,)
    assert np.all(out * 32768 == np.round(out * 32768))
    s = OS.time_stretch(np.float32(pcm) / np.float32(32767), 0.9)[-16000:]
    q = np.trunc(np.float32(s) * np.float32(32767))           # np.int16() truncates toward zero
    assert np.array_equal(out * 32768, q)
    short = OS.tta_slow_clip(pcm[:12000])
    n = OS.stretched_length(12000, 0.9)
    assert n < 16000 and np.all(short[n:] == 0) and np.any(short[:n] != 0)
    assert np.all(OS.tta_slow_clip(np.zeros(16000, np.int16)) == 0)
