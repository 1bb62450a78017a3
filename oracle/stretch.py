"""Oracle: the speed-TTA time stretch.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

The reference builds its slow test set offline (create_tta_set.py:9-22):

    data = np.float32(int16_wav) / 32767
    data = librosa.effects.time_stretch(data, 0.9)      # third-party, not under /root/reference
    data = data[-16000:]
    wavfile.write(out, rate, np.int16(data * 32767))

and make_submission.py:86-100,133-136 reads those files back through the DecodeWav graph
(int16 / 32768).  librosa is not pinned by the reference (README lists only TF and Keras) and is not
installed here, so this file restates the PUBLISHED algorithm of librosa 0.5.x (the release current
when the reference was written, Jan 2018) - PARITY UNPINNED:

    effects.time_stretch(y, rate) = core.istft(core.phase_vocoder(core.stft(y), rate), dtype=y.dtype)

  * core.stft: n_fft 2048, hop n_fft//4 = 512, periodic Hann (scipy get_window('hann', fftbins=True)),
    center=True -> np.pad(y, n_fft//2, mode='reflect'); frames y[512 i : 512 i + 2048] * window; rfft;
    1 + len(y)//512 frames (0.5.x stores the conjugate of the rfft and istft conjugates back - the
    convention cancels in time_stretch, see phase_vocoder below).
  * core.phase_vocoder: time_steps = np.arange(0, n_frames, rate); D padded with two zero columns;
    per step: linear interpolation of the magnitudes of columns int(step), int(step)+1 with
    alpha = step mod 1; phase accumulator starts at angle(D[:, 0]) and advances by
    phi_advance + princarg(angle(c1) - angle(c0) - phi_advance), phi_advance = linspace(0, pi*hop, 1025).
    The accumulator is only ever used as exp(1j*acc), so it is angle(c1) - angle(c0) modulo 2*pi whatever
    phi_advance is - which is also why the sign convention of the STFT does not matter.
  * core.istft: per frame window * irfft(column), overlap-add at 512 t, divide by the window
    sum-of-squares (window_sumsquare) where that exceeds tiny, then trim n_fft//2 from both ends
    (no `length` argument in 0.5.x/0.6.x: the result has 512*(n_out_frames-1) samples; librosa >= 0.7
    passes length=round(len(y)/rate) instead - `length_mode='round'` restates that variant).

`dtype=np.float64` (default) gives the value both librosa's complex64/float32 arithmetic and the HIP
f32 kernel approximate; `literal_f32=True` follows librosa 0.5.x's own dtypes (complex64 spectra,
float32 accumulator that grows to ~6e4 rad) and is used by the tests only to show how much noise the
reference's own rounding carries.
"""
import numpy as np

N_FFT = 2048
HOP = 512


def hann_periodic(n=N_FFT):
    """scipy.signal.get_window('hann', n, fftbins=True)."""
    return 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n) / n)


def stft(y, dtype=np.float64):
    """librosa.core.stft(y) with its defaults -> [1025, n_frames] complex."""
    y = np.asarray(y, dtype=dtype)
    yp = np.pad(y, N_FFT // 2, mode='reflect')
    n_frames = 1 + (len(yp) - N_FFT) // HOP
    win = hann_periodic().astype(dtype)
    frames = np.stack([yp[HOP * i:HOP * i + N_FFT] * win for i in range(n_frames)], axis=1)
    D = np.fft.rfft(frames, axis=0)
    return D.astype(np.complex64 if dtype == np.float32 else np.complex128)


def time_steps(n_frames, rate):
    """np.arange(0, n_frames, rate, dtype=float) -> (int(step), step mod 1.0) per output frame."""
    steps = np.arange(0, n_frames, rate, dtype=np.float64)
    return steps.astype(np.int64), np.mod(steps, 1.0)


def phase_vocoder(D, rate, literal_f32=False):
    """librosa.core.phase_vocoder(D, rate, hop_length=512)."""
    n_bins, n_frames = D.shape
    idx, alpha = time_steps(n_frames, rate)
    out = np.zeros((n_bins, len(idx)), dtype=D.dtype)
    phi_advance = np.linspace(0, np.pi * HOP, n_bins)
    acc = np.angle(D[:, 0])
    if literal_f32:
        acc = acc.astype(np.float32)
    Dp = np.pad(D, [(0, 0), (0, 2)], mode='constant')
    for t in range(len(idx)):
        c0, c1 = Dp[:, idx[t]], Dp[:, idx[t] + 1]
        mag = (1.0 - alpha[t]) * np.abs(c0) + alpha[t] * np.abs(c1)
        out[:, t] = mag * np.exp(1.j * acc)
        dphase = np.angle(c1) - np.angle(c0) - phi_advance
        dphase = dphase - 2.0 * np.pi * np.round(dphase / (2.0 * np.pi))
        if literal_f32:
            acc = (acc + (phi_advance + dphase)).astype(np.float32)
        else:
            acc = acc + phi_advance + dphase
    return out


def window_sumsquare(n_frames, dtype=np.float64):
    """librosa.filters.window_sumsquare('hann', n_frames, hop 512, n_fft 2048, norm=None)."""
    n = N_FFT + HOP * (n_frames - 1)
    x = np.zeros(n, dtype=dtype)
    w2 = hann_periodic().astype(dtype) ** 2
    for i in range(n_frames):
        x[HOP * i:HOP * i + N_FFT] += w2
    return x


def istft(D, dtype=np.float64, length=None):
    """librosa.core.istft(D) with its defaults (center=True)."""
    n_frames = D.shape[1]
    n = N_FFT + HOP * (n_frames - 1)
    y = np.zeros(n, dtype=dtype)
    win = hann_periodic().astype(dtype)
    for i in range(n_frames):
        y[HOP * i:HOP * i + N_FFT] += win * np.fft.irfft(D[:, i], N_FFT).astype(dtype)
    ss = window_sumsquare(n_frames, dtype)
    nz = ss > np.finfo(dtype).tiny
    y[nz] /= ss[nz]
    if length is None:
        return y[N_FFT // 2:-(N_FFT // 2)]
    y = y[N_FFT // 2:]
    if len(y) >= length:
        return y[:length]
    return np.pad(y, (0, length - len(y)), mode='constant')


def time_stretch(y, rate, dtype=np.float64, literal_f32=False, length_mode='trim'):
    """librosa.effects.time_stretch(y, rate).  length_mode 'trim' = librosa 0.5/0.6 (istft without a
    length), 'round' = librosa >= 0.7 (length = round(len(y)/rate))."""
    if rate <= 0:
        raise ValueError('rate must be a positive number')
    if literal_f32:
        dtype = np.float32
    D = phase_vocoder(stft(y, dtype), rate, literal_f32)
    length = None if length_mode == 'trim' else int(round(len(y) / rate))
    return istft(D, dtype, length)


def stretched_length(n_samples, rate):
    """Samples librosa 0.5's time_stretch returns for an n_samples input."""
    n_frames = 1 + n_samples // HOP
    return HOP * (len(np.arange(0, n_frames, rate)) - 1)


def tta_slow_clip(pcm_int16, rate=0.9, keep=16000, dtype=np.float64, literal_f32=False):
    """create_tta_set.py:15-22 followed by the DecodeWav read of make_submission.py:86-100: the slow clip
    as the model sees it, float32 [keep] (zero padded at the end when the stretched signal is shorter,
    input_data.py:335-336 desired_samples)."""
    data = np.float32(np.asarray(pcm_int16)) / np.float32(32767)
    data = time_stretch(data, rate, dtype, literal_f32)[-keep:]
    q = np.int16(np.float32(data) * np.float32(32767))         # np.int16() truncates toward zero
    out = np.zeros(keep, np.float32)
    out[:len(q)] = q.astype(np.float32) / np.float32(32768.0)
    return out
