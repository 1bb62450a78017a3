"""librosa.effects.time_stretch for the reference's create_tta_set.py:19, computed by `kws_time_stretch_f32`
(one clip per call, as the script makes it; `speech_recognition_amd.tta.time_stretch` is the batched form)."""
import numpy as np

from speech_recognition_amd import tta as _tta


def time_stretch(y, rate):
    """Returns the whole stretched signal (librosa 0.5.x length: 512 * (n_output_frames - 1)) as float32."""
    y = np.ascontiguousarray(np.asarray(y, dtype=np.float32).reshape(-1))
    if rate <= 0:
        raise ValueError('rate must be a positive number')
    n = _tta.stretched_samples(len(y), rate)
    out = _tta.time_stretch(y.reshape(1, -1), rate, keep=n, wav_round_trip=False, in_scale=1.0)
    return out.cpu().numpy().reshape(-1)
