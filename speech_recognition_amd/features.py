"""Host-side table builders for the table-driven STFT/mel/DCT kernel (kws_stft_mel_f32).

The tables (window, mel weight matrix, DCT matrix, log offset/floor) are the only place where the
two reference feature paths differ; they are computed once on the host the way the TF-1.4 ops
compute them and uploaded into a `kws_stft_plan`.
  path B: tf.contrib.signal.{hann_window, linear_to_mel_weight_matrix, mfccs_from_log_mel_spectrograms}
          as called at reference input_data.py:361-381 (float32 in-graph)
  path A: contrib_audio.{audio_spectrogram, mfcc} as called at reference audio.py:15-23 (double)
"""
import numpy as np

F32 = np.float32


def _pow2_at_least(n):
    p = 1
    while p < n:
        p <<= 1
    return p


def _mel_f32(hz):
    return F32(1127.0) * np.log(F32(1.0) + np.asarray(hz, dtype=F32) / F32(700.0))


def path_b_tables(window_size, num_mel_bins, num_keep, sample_rate=16000, lower_hz=80.0, upper_hz=7600.0):
    fft_len = _pow2_at_least(window_size)
    n_bins = fft_len // 2 + 1
    # periodic Hann: 0.5 - 0.5 cos(2 pi i / N) for even N
    denom = F32(window_size + (1 - window_size % 2) - 1)
    win = (F32(0.5) - F32(0.5) * np.cos(F32(2.0 * np.pi) * np.arange(window_size, dtype=F32) / denom)).astype(F32)
    # HTK-style triangular filters on the mel scale, DC bin excluded
    bins_mel = _mel_f32(np.linspace(F32(0.0), F32(sample_rate / 2.0), n_bins).astype(F32)[1:])[:, None]
    edges = np.linspace(_mel_f32(lower_hz), _mel_f32(upper_hz), num_mel_bins + 2).astype(F32)
    lo, ce, hi = edges[None, :-2], edges[None, 1:-1], edges[None, 2:]
    w = np.maximum(F32(0.0), np.minimum((bins_mel - lo) / (ce - lo), (hi - bins_mel) / (hi - ce))).astype(F32)
    mel = np.ascontiguousarray(np.pad(w, [[1, 0], [0, 0]]), dtype=F32)
    # DCT-II scaled by rsqrt(2M), first num_keep coefficients: D[m, q]
    m = np.arange(num_mel_bins, dtype=np.float64)[:, None]
    q = np.arange(num_keep, dtype=np.float64)[None, :]
    dct = (2.0 * np.cos(np.pi * q * (2.0 * m + 1.0) / (2.0 * num_mel_bins)) / np.sqrt(2.0 * num_mel_bins))
    return dict(window=np.ascontiguousarray(win), fft_length=fft_len, mel=mel,
                dct=np.ascontiguousarray(dct, dtype=F32), log_offset=1e-6, log_floor=0.0)


def path_a_tables(window_size=480, sample_rate=16000, dct_coefficient_count=40, filterbank_channel_count=40,
                  lower=20.0, upper=4000.0):
    fft_len = _pow2_at_least(window_size)
    n_bins = fft_len // 2 + 1
    i = np.arange(window_size, dtype=np.float64)
    win = 0.5 - 0.5 * np.cos(2.0 * np.pi * i / window_size)

    def f2m(f):
        return 1127.0 * np.log(1.0 + f / 700.0)
    nch = filterbank_channel_count
    mel_lo, mel_hi = f2m(lower), f2m(upper)
    centers = mel_lo + (mel_hi - mel_lo) / (nch + 1) * (np.arange(nch + 1) + 1)
    hz_per_bin = 0.5 * sample_rate / (n_bins - 1)
    start, end = int(1.5 + lower / hz_per_bin), int(upper / hz_per_bin)
    mel = np.zeros((n_bins, nch))
    ch = 0
    for k in range(start, min(end, n_bins - 1) + 1):
        mk = f2m(k * hz_per_bin)
        while ch < nch and centers[ch] < mk:
            ch += 1
        c = ch - 1
        left, right = (centers[c], centers[c + 1]) if c >= 0 else (mel_lo, centers[0])
        wt = (right - mk) / (right - left)
        if c >= 0:
            mel[k, c] += wt
        if c + 1 < nch:
            mel[k, c + 1] += 1.0 - wt
    jj = np.arange(nch)[:, None]
    ii = np.arange(dct_coefficient_count)[None, :]
    dct = np.sqrt(2.0 / nch) * np.cos(ii * (np.pi / nch) * (jj + 0.5))
    return dict(window=np.ascontiguousarray(win, dtype=F32), fft_length=fft_len,
                mel=np.ascontiguousarray(mel, dtype=F32), dct=np.ascontiguousarray(dct, dtype=F32),
                log_offset=0.0, log_floor=1e-12)
