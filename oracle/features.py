"""Oracle: augmentation graph + STFT / mel / log / DCT feature paths (rows a2-a5).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Restates, in NumPy, what the reference's TF-1.4 feature graph computes:
  * path B - reference input_data.py:331-381 (tf.contrib.signal.stft ->
    tf.abs -> linear_to_mel_weight_matrix -> tf.log(+1e-6) ->
    mfccs_from_log_mel_spectrograms[..., :K]), SURVEY.md Appendix A.1;
  * path A - reference audio.py:6-23 (contrib_audio.audio_spectrogram ->
    contrib_audio.mfcc), SURVEY.md Appendix A.2.
Both paths are expressed through the same four tables (window, mel matrix,
log offset/floor, DCT matrix), which is also how the HIP kernel is driven.

``dtype`` selects the arithmetic type of the restatement: float64 (default)
gives the value both TF-fp32 and the HIP-fp32 kernel approximate; float32
mimics the reference's own rounding more closely.  Tables are always built
the way TF builds them (float32 for path B, double for path A).
"""
import numpy as np

# ----------------------------------------------------------------------------
# a2: augment  (reference input_data.py:334-359, utils.py:56-73)
# ----------------------------------------------------------------------------


def tf_roll(a, shift):
    """utils.py:56-73 - circular shift along axis 0: both branches of the
    tf.cond equal np.roll(a, shift) (shift>=0: concat(a[L-s:], a[:L-s]);
    shift<0: concat(a[-s:], a[:-s]))."""
    return np.roll(a, int(shift), axis=0)


def augment(clip, fg_volume, time_shift, background, bg_volume, dtype=np.float32):
    """input_data.py:340-359 - y[t] = bg[t]*bg_volume + roll(clip*fg, shift)[t].

    TF evaluates in float32: multiply, roll, multiply, add (background_mul +
    shifted_foreground, input_data.py:353-355).  No clipping (:356)."""
    clip = np.asarray(clip, dtype=dtype)
    scaled = clip * dtype(fg_volume)
    shifted = np.roll(scaled, int(time_shift))
    bg = np.asarray(background, dtype=dtype) * dtype(bg_volume)
    return (bg + shifted).astype(dtype)


def augment_batch(bank, clip_idx, fg_volume, time_shift, noise, noise_off, bg_volume,
                  dtype=np.float32):
    """Batched form used by the tests: bank [N, L], noise = 1-D concatenation of
    the background recordings, noise_off = absolute start of each 1 s slice
    (input_data.py:482-488)."""
    B = len(clip_idx)
    L = bank.shape[1]
    out = np.empty((B, L), dtype=dtype)
    for b in range(B):
        seg = noise[noise_off[b]:noise_off[b] + L]
        out[b] = augment(bank[clip_idx[b]], fg_volume[b], time_shift[b], seg, bg_volume[b], dtype)
    return out


# ----------------------------------------------------------------------------
# tables
# ----------------------------------------------------------------------------


def enclosing_power_of_two(n):
    """tf.contrib.signal.stft fft_length=None -> smallest power of two >= frame_length."""
    p = 1
    while p < n:
        p *= 2
    return p


def hann_periodic(n, dtype=np.float32):
    """tf.contrib.signal.hann_window(periodic=True): 0.5 - 0.5*cos(2*pi*i/n')
    with n' = n for even n (window_ops._raised_cosine_window); SURVEY D.1."""
    even = 1 - n % 2
    denom = dtype(n + even - 1)
    count = np.arange(n, dtype=dtype)
    cos_arg = dtype(2.0 * np.pi) * count / denom
    return (dtype(0.5) - dtype(0.5) * np.cos(cos_arg)).astype(dtype)


def hertz_to_mel(f, dtype=np.float32):
    return dtype(1127.0) * np.log(dtype(1.0) + np.asarray(f, dtype=dtype) / dtype(700.0))


def linear_to_mel_weight_matrix(num_mel_bins, num_spectrogram_bins, sample_rate,
                                lower_edge_hertz, upper_edge_hertz, dtype=np.float32):
    """tf.contrib.signal.linear_to_mel_weight_matrix (TF 1.4), float32 in-graph;
    called at reference input_data.py:369-373 with (M, 257, 16000, 80, 7600)."""
    nyquist = dtype(sample_rate / 2.0)
    lin = np.linspace(dtype(0.0), nyquist, num_spectrogram_bins).astype(dtype)[1:]
    spec_mel = hertz_to_mel(lin, dtype)[:, None]
    edges = np.linspace(hertz_to_mel(lower_edge_hertz, dtype), hertz_to_mel(upper_edge_hertz, dtype),
                        num_mel_bins + 2).astype(dtype)
    lower = edges[None, :-2]
    center = edges[None, 1:-1]
    upper = edges[None, 2:]
    lower_slopes = (spec_mel - lower) / (center - lower)
    upper_slopes = (upper - spec_mel) / (upper - center)
    w = np.maximum(dtype(0.0), np.minimum(lower_slopes, upper_slopes)).astype(dtype)
    return np.pad(w, [[1, 0], [0, 0]])


def dct2_matrix(num_mel_bins, num_keep, dtype=np.float64):
    """mfccs_from_log_mel_spectrograms (TF 1.4): dct2(x)[q] * rsqrt(2M) with
    dct2(x)[q] = Re(2 e^{-j pi q / 2M} * rfft(x, 2M)[q]) = 2 sum_m x[m] cos(pi q (2m+1)/(2M)).
    Returned as D[m, q] so that mfcc = L @ D.  Constants pinned by the graph_def
    (Rsqrt/x=160, dct/mul_1/x=2, dct/mul/x=-pi)."""
    m = np.arange(num_mel_bins, dtype=np.float64)[:, None]
    q = np.arange(num_keep, dtype=np.float64)[None, :]
    d = 2.0 * np.cos(np.pi * q * (2.0 * m + 1.0) / (2.0 * num_mel_bins)) / np.sqrt(2.0 * num_mel_bins)
    return d.astype(dtype)


def tables_path_b(window_size=480, num_mel_bins=80, num_keep=60, sample_rate=16000,
                  lower_hz=80.0, upper_hz=7600.0):
    """Tables of reference input_data.py:360-381 (path B)."""
    fft_len = enclosing_power_of_two(window_size)
    return dict(
        window=hann_periodic(window_size, np.float32),
        fft_length=fft_len,
        mel=linear_to_mel_weight_matrix(num_mel_bins, fft_len // 2 + 1, sample_rate,
                                        lower_hz, upper_hz, np.float32),
        log_offset=1e-6, log_floor=0.0,
        dct=dct2_matrix(num_mel_bins, num_keep, np.float64).astype(np.float32),
        power=False,
    )


def mfcc_mel_filterbank_dense(input_length=257, sample_rate=16000.0, num_channels=40,
                              lower=20.0, upper=4000.0):
    """TF 1.4 core/kernels/mfcc_mel_filterbank.cc, written as a dense [bins, channels]
    matrix so that out = sqrt(power) @ W."""
    def f2m(f):
        return 1127.0 * np.log(1.0 + f / 700.0)
    mel_low, mel_hi = f2m(lower), f2m(upper)
    spacing = (mel_hi - mel_low) / (num_channels + 1)
    centers = np.array([mel_low + spacing * (i + 1) for i in range(num_channels + 1)])
    hz_per_sbin = 0.5 * sample_rate / (input_length - 1)
    start = int(1.5 + lower / hz_per_sbin)
    end = int(upper / hz_per_sbin)
    w = np.zeros((input_length, num_channels), dtype=np.float64)
    channel = 0
    for i in range(input_length):
        if i < start or i > end:
            continue
        melf = f2m(i * hz_per_sbin)
        while channel < num_channels and centers[channel] < melf:
            channel += 1
        c = channel - 1
        if c >= 0:
            wt = (centers[c + 1] - melf) / (centers[c + 1] - centers[c])
        else:
            wt = (centers[0] - melf) / (centers[0] - mel_low)
        if c >= 0:
            w[i, c] += wt
        if c + 1 < num_channels:
            w[i, c + 1] += 1.0 - wt
    return w


def mfcc_dct_matrix(input_length=40, coefficient_count=40):
    """TF 1.4 core/kernels/mfcc_dct.cc: cos[i][j] = sqrt(2/N) cos(i*pi/N*(j+0.5)); D[j, i]."""
    fnorm = np.sqrt(2.0 / input_length)
    arg = np.pi / input_length
    i = np.arange(coefficient_count)[None, :]
    j = np.arange(input_length)[:, None]
    return fnorm * np.cos(i * arg * (j + 0.5))


def tables_path_a(window_size=480, sample_rate=16000, dct_coefficient_count=40,
                  filterbank_channel_count=40, lower=20.0, upper=4000.0):
    """Tables of reference audio.py:15-23 (audio_spectrogram magnitude_squared=True -> mfcc)."""
    fft_len = enclosing_power_of_two(window_size)
    i = np.arange(window_size, dtype=np.float64)
    return dict(
        window=(0.5 - 0.5 * np.cos(2.0 * np.pi * i / window_size)),   # spectrogram.cc GetPeriodicHann (double)
        fft_length=fft_len,
        mel=mfcc_mel_filterbank_dense(fft_len // 2 + 1, float(sample_rate), filterbank_channel_count,
                                      lower, upper),
        log_offset=0.0, log_floor=1e-12,                            # mfcc.cc kFilterbankFloor
        dct=mfcc_dct_matrix(filterbank_channel_count, dct_coefficient_count),
        power=True,
    )


# ----------------------------------------------------------------------------
# a3 + a4 (+ a5): STFT -> |X| -> mel -> log -> DCT
# ----------------------------------------------------------------------------


def frame_signal(x, frame_length, frame_step):
    """tf.contrib.signal.frame(pad_end=False): F = 1 + (L - frame_length)//frame_step."""
    L = x.shape[-1]
    nf = 1 + (L - frame_length) // frame_step if L >= frame_length else 0
    idx = np.arange(frame_length)[None, :] + frame_step * np.arange(nf)[:, None]
    return x[..., idx]


def stft_magnitude(x, tables, frame_step=160, dtype=np.float64):
    """input_data.py:361-366: |rfft(frame * hann, 512)|  -> [..., F, 257]."""
    win = np.asarray(tables['window'], dtype=dtype)
    frames = frame_signal(np.asarray(x, dtype=dtype), len(win), frame_step) * win
    spec = np.fft.rfft(frames.astype(np.float64), n=tables['fft_length'], axis=-1)
    mag = np.abs(spec)
    return mag.astype(dtype)


def features(x, tables, frame_step=160, dtype=np.float64, return_all=False):
    """Full path: [..., L] waveform -> [..., F, K] 'mfcc_' tensor.

    path B (input_data.py:366-381): log(|X| @ W + 1e-6) @ D
    path A (audio.py:15-23):        log(max(sqrt(|X|^2) @ W, 1e-12)) @ D
    """
    mag = stft_magnitude(x, tables, frame_step, dtype)
    lead = mag.shape[:-1]
    # the two products run on [frames, bins] matrices (one GEMM each; a stacked matmul loops over the batch)
    mel = mag.reshape(-1, mag.shape[-1]) @ np.asarray(tables['mel'], dtype=dtype)
    mel = mel + dtype(tables['log_offset'])
    if tables['log_floor'] > 0.0:
        mel = np.maximum(mel, dtype(tables['log_floor']))
    logmel = np.log(mel)
    out = (logmel @ np.asarray(tables['dct'], dtype=dtype)).reshape(lead + (-1,))
    logmel = logmel.reshape(lead + (-1,))
    if return_all:
        return mag, logmel, out
    return out


def features_batched(x, tables, frame_step=160, workers=-1, chunk=16):
    """The same path B / path A features the way a CPU production path batches them ("B2"): float32
    throughout like TF's CPU kernels, scipy.fft.rfft over a [clips * F, 512] frame matrix, mel and DCT as ONE GEMM each per
    chunk - in cache-sized chunks of `chunk` clips dealt to `workers` threads (NumPy / pocketfft release the GIL).  One
    whole-batch rfft(workers=-1) was measured too: its 150 MB of intermediates per 256 clips make it SLOWER per clip than
    the per-clip loop (0.58 against 0.26 ms on one thread).  Checked against features() (float64) in
    tests/test_oracle_crosscheck_cpu.py."""
    import os
    import scipy.fft
    from concurrent.futures import ThreadPoolExecutor
    win = np.asarray(tables['window'], dtype=np.float32)
    mel = np.asarray(tables['mel'], dtype=np.float32)
    dct = np.asarray(tables['dct'], dtype=np.float32)
    x = np.ascontiguousarray(x, dtype=np.float32)
    lead = x.shape[:-1]
    x2 = x.reshape(-1, x.shape[-1])

    def one(xc):
        # frames as a strided VIEW of the clips, windowed straight into one contiguous [clips * F, frame_length] matrix
        view = np.lib.stride_tricks.sliding_window_view(xc, len(win), axis=-1)[..., ::frame_step, :]
        frames = np.empty(view.shape, np.float32)
        np.multiply(view, win, out=frames)
        mag = np.abs(scipy.fft.rfft(frames.reshape(-1, len(win)), n=tables['fft_length'], axis=-1))   # complex64 -> f32
        m = mag @ mel
        m += np.float32(tables['log_offset'])
        if tables['log_floor'] > 0.0:
            np.maximum(m, np.float32(tables['log_floor']), out=m)
        np.log(m, out=m)
        return (m @ dct).reshape(xc.shape[0], view.shape[-2], -1)

    chunks = [x2[i:i + chunk] for i in range(0, x2.shape[0], chunk)]
    n_thr = (os.cpu_count() or 1) if workers is None or workers < 1 else int(workers)
    if n_thr <= 1 or len(chunks) == 1:
        parts = [one(c) for c in chunks]
    else:
        with ThreadPoolExecutor(min(n_thr, len(chunks))) as ex:
            parts = list(ex.map(one, chunks))
    out = np.concatenate(parts, axis=0)
    return out.reshape(lead + out.shape[1:])


def features_per_clip_f64(x, tables, frame_step=160):
    """'Reference-style' driver used by the CPU baseline: one clip per call, result
    copied into a float64 row, like the per-clip sess.run loop of
    input_data.py:457-536."""
    out = None
    for i in range(x.shape[0]):
        f = features(x[i], tables, frame_step, dtype=np.float32).reshape(-1)
        if out is None:
            out = np.zeros((x.shape[0], f.size))
        out[i, :] = f
    return out
