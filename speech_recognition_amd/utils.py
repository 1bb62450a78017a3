"""Host-side mirror of the reference's utils.py: the batch generator around AudioProcessor.get_data
(reference utils.py:6-53), the circular roll (utils.py:56-73), center_crop (utils.py:76-84) and the
loss used by the time-sliced attention model (utils.py:87-108; on the device it is fused into the
network tail, this function is the host-callable form for metric code)."""
from __future__ import division, print_function

import numpy as np
import torch


def data_gen(audio_processor, sess, batch_size=128, background_frequency=0.3, background_volume_range=0.15,
             foreground_frequency=0.3, foreground_volume_range=0.15, time_shift_frequency=0.3,
             time_shift_range=[-500, 0], mode='validation', pseudo_frequency=0.33, flip_frequency=0.0,
             silence_volume_range=0.3):
    """Infinite generator of (X, y).  Non-training modes switch every augmentation off except
    silence_volume_range and walk the partition in order, wrapping before the last partial batch
    (reference utils.py:15-40)."""
    ep_count = 0
    offset = 0
    if mode != 'training':
        background_frequency = background_volume_range = 0.0
        foreground_frequency = foreground_volume_range = 0.0
        pseudo_frequency = time_shift_frequency = flip_frequency = 0.0
        time_shift_range = [0, 0]
    while True:
        X, y = audio_processor.get_data(
            how_many=batch_size, offset=0 if mode == 'training' else offset,
            background_frequency=background_frequency, background_volume_range=background_volume_range,
            foreground_frequency=foreground_frequency, foreground_volume_range=foreground_volume_range,
            time_shift_frequency=time_shift_frequency, time_shift_range=time_shift_range, mode=mode, sess=sess,
            pseudo_frequency=pseudo_frequency, flip_frequency=flip_frequency,
            silence_volume_range=silence_volume_range)
        offset += batch_size
        if offset > audio_processor.set_size(mode) - batch_size:
            offset = 0
            print("\n[Ep:%03d: %s-mode]: Pseudo: %.3f" % (ep_count, mode, pseudo_frequency))
            ep_count += 1
        yield X, y


def pseudo_schedule(ep_count):
    """The pseudo-label mix schedule the reference keeps as a comment (utils.py:41-49):
    1.0 for epochs <= 20, 0.7 <= 30, 0.4 <= 40, else 0.2."""
    if ep_count <= 20:
        return 1.0
    if ep_count <= 30:
        return 0.7
    if ep_count <= 40:
        return 0.4
    return 0.2


def tf_roll(a, shift, a_len=16000):
    """Circular shift along axis 0 (= np.roll(a, shift, axis=0)); accepts torch tensors or arrays."""
    if torch.is_tensor(a):
        return torch.roll(a, int(shift), dims=0)
    return np.roll(a, int(shift), axis=0)


def center_crop(data, desired_size=16000):
    if data.ndim == 1:
        left = (len(data) - desired_size) // 2
        return data[left: left + desired_size]
    if data.ndim == 2:
        left = (data.shape[1] - desired_size) // 2
        return data[:, left: left + desired_size]
    raise RuntimeError("Invalid tensor shape: %s" % (list(data.shape)))


def smooth_categorical_crossentropy(target, output, from_logits=False, label_smoothing=0.0):
    """Mean softmax cross-entropy between smoothed targets and log(clip(output, 1e-7, 1-1e-7))."""
    t = np.asarray(target, dtype=np.float64)
    z = np.asarray(output, dtype=np.float64)
    if not from_logits:
        z = np.log(np.clip(z, 1e-7, 1.0 - 1e-7))
    t = t * (1.0 - label_smoothing) + label_smoothing / t.shape[-1]
    z = z - z.max(axis=-1, keepdims=True)
    logp = z - np.log(np.exp(z).sum(axis=-1, keepdims=True))
    return float((-(t * logp).sum(axis=-1)).mean())
