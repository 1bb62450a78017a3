"""AudioConverter (reference audio.py:6-28): decode_wav -> AudioSpectrogram(480,160,squared) ->
Mfcc(40 channels, 20..4000 Hz, 40 coefficients), evaluated by the same table-driven HIP kernel as
the training features (kws_stft_mel_f32) with the path-A tables."""
import ctypes

import numpy as np
import torch

from . import _lib
from .features import path_a_tables
from .input_data import _read_wav_int16


class AudioConverter(object):
    def __init__(self, desired_samples=16000, window_size_samples=480, window_stride_samples=160, device=None):
        if not torch.cuda.is_available():
            raise _lib.KwsError("AudioConverter needs an MI355X (no CPU fallback)")
        self.lib = _lib.load()
        self.device = torch.device(device if device is not None else "cuda:%d" % torch.cuda.current_device())
        self.desired_samples = desired_samples
        t = path_a_tables(window_size_samples, 16000, 40, 40)
        plan = ctypes.c_void_p()
        _lib.check(self.lib.kws_stft_plan_create(
            window_size_samples, window_stride_samples, t['fft_length'], 40, 40,
            t['window'].ctypes.data_as(ctypes.c_void_p), t['mel'].ctypes.data_as(ctypes.c_void_p),
            t['dct'].ctypes.data_as(ctypes.c_void_p), t['log_offset'], t['log_floor'], ctypes.byref(plan)),
            "kws_stft_plan_create")
        self._plan = plan
        self._frames = self.lib.kws_stft_num_frames(plan, desired_samples)

    def __del__(self):
        try:
            if getattr(self, '_plan', None):
                self.lib.kws_stft_plan_destroy(self._plan)
                self._plan = None
        except Exception:
            pass

    def convert(self, x):
        """x: f32 CUDA tensor [B, desired_samples] -> [B, frames, 40] on the device."""
        B = x.shape[0]
        out = torch.empty((B, self._frames, 40), dtype=torch.float32, device=self.device)
        _lib.call("kws_stft_mel_f32", self._plan, _lib.ptr(x), B, self.desired_samples, _lib.ptr(out), 0,
                  _lib.stream_ptr())
        return out

    def load(self, fn, sess=None):
        a, _ = _read_wav_int16(fn)
        buf = np.zeros((1, self.desired_samples), np.float32)
        n = min(len(a), self.desired_samples)
        buf[0, :n] = a[:n].astype(np.float32) / np.float32(32768.0)
        return self.convert(torch.from_numpy(buf).to(self.device)).cpu().numpy()
