"""Oracle: control logic of the data path (row a1/a6) - SHA-1 partition, index construction,
the per-clip RNG draw order of AudioProcessor.get_data, data_gen's offset logic, the Keras
ReduceLROnPlateau rule and prepare_model_settings.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Restates reference input_data.py:61-114,
182-272, 395-541, utils.py:6-53, model.py:1785-1829 and SURVEY.md Appendix C / D.6.  PINNED: every
function here is checked against fixtures captured from the reference's own modules
(tests/golden/k5_control_logic.json, k3_scalars.json; generator tests/golden/make_golden.py).
"""
import hashlib
import math
import os
import random
import re

import numpy as np

MAX_NUM_WAVS_PER_CLASS = 2 ** 27 - 1
SILENCE_LABEL = '_silence_'
UNKNOWN_WORD_INDEX = 1
RANDOM_SEED = 59185


def which_set(filename, validation_percentage, testing_percentage):
    """input_data.py:61-114"""
    if os.path.basename(os.path.dirname(filename)) == 'unknown_unknown':
        return 'training'
    base = os.path.basename(filename)
    if '_nohash_' not in base:
        return 'pseudo'
    h = hashlib.sha1(re.sub(r'_nohash_.*$', '', base).encode()).hexdigest()
    pct = (int(h, 16) % (MAX_NUM_WAVS_PER_CLASS + 1)) * (100.0 / MAX_NUM_WAVS_PER_CLASS)
    if pct < validation_percentage:
        return 'validation'
    if pct < testing_percentage + validation_percentage:
        return 'testing'
    return 'training'


def build_index(wav_paths, wanted_words, silence_percentage, unknown_percentage, validation_percentage,
                testing_percentage):
    """input_data.py:182-272 for ONE data dir whose sorted glob is `wav_paths`.
    Returns (data_index {set: [(label, file)]}, word_to_index)."""
    random.seed(RANDOM_SEED)
    wanted = {w: i + 2 for i, w in enumerate(wanted_words)}
    sets = ['validation', 'testing', 'training', 'pseudo']
    index = {s: [] for s in sets}
    unknown = {s: [] for s in sets}
    all_words = {}
    for p in wav_paths:
        word = re.search('.*/([^/]+)/.*.wav', p).group(1).lower()
        if word == '_background_noise_':
            continue
        all_words[word] = True
        s = which_set(p, validation_percentage, testing_percentage)
        (index if word in wanted else unknown)[s].append((word, p))
    silence_file = index['training'][0][1]
    for s in sets:
        n = len(index[s])
        index[s].extend([(SILENCE_LABEL, silence_file)] * int(math.ceil(n * silence_percentage / 100)))
        random.shuffle(unknown[s])
        index[s].extend(unknown[s][:int(math.ceil(n * unknown_percentage / 100))])
    for s in sets:
        random.shuffle(index[s])
    word_to_index = {w: wanted.get(w, UNKNOWN_WORD_INDEX) for w in all_words}
    word_to_index[SILENCE_LABEL] = 0
    return index, word_to_index


def draw_batch(index, mode, how_many, offset, n_background, background_len, desired_samples,
               background_frequency, background_volume_range, foreground_frequency, foreground_volume_range,
               time_shift_frequency, time_shift_range, pseudo_frequency=0.0, flip_frequency=0.0,
               silence_volume_range=0.0):
    """input_data.py:428-514: per clip, in this order - sample pick, time shift, background
    (index, offset, volume), foreground volume / flip.  Uses the NumPy GLOBAL RNG like the reference.
    Returns a list of dicts (file, label, time_shift, bg_index, bg_offset, bg_volume, fg_volume)."""
    cand, pseudo = index[mode], index['pseudo']
    count = len(cand) if how_many == -1 else max(0, min(how_many, len(cand) - offset))
    use_bg = n_background > 0 and mode == 'training'
    out = []
    for i in range(offset, offset + count):
        if how_many == -1 or mode != 'training':
            label, fn = cand[i]
        elif np.random.uniform(0, 1) < pseudo_frequency:
            label, fn = pseudo[np.random.randint(len(pseudo))]
        else:
            label, fn = cand[np.random.randint(len(cand))]
        shift = 0
        if np.random.uniform(0.0, 1.0) < time_shift_frequency:
            shift = np.random.randint(time_shift_range[0], time_shift_range[1] + 1)
        bg_index, bg_offset, bg_volume = -1, 0, 0.0
        if use_bg:
            bg_index = np.random.randint(n_background)
            bg_offset = np.random.randint(0, background_len[bg_index] - desired_samples)
            if np.random.uniform(0, 1) < background_frequency:
                bg_volume = np.random.uniform(0, background_volume_range)
            elif label == SILENCE_LABEL and np.random.uniform(0, 1) < 0.9:
                bg_volume = np.random.uniform(0, silence_volume_range)
        if label == SILENCE_LABEL:
            fg = 0.0
        else:
            fg = 1.0
            if np.random.uniform(0, 1) < foreground_frequency:
                fg = 1.0 + np.random.uniform(-foreground_volume_range, foreground_volume_range)
            if np.random.uniform(0, 1) < flip_frequency:
                fg *= -1.0
        out.append(dict(file=fn, label=label, time_shift=int(shift), bg_index=int(bg_index),
                        bg_offset=int(bg_offset), bg_volume=float(bg_volume), fg_volume=float(fg)))
    return out


def data_gen_plan(set_size, batch_size, mode, n_batches):
    """utils.py:25-53: the (offset, how_many) sequence data_gen feeds to get_data."""
    offset, plan = 0, []
    for _ in range(n_batches):
        plan.append((0 if mode == 'training' else offset, batch_size))
        offset += batch_size
        if offset > set_size - batch_size:
            offset = 0
    return plan


def reduce_lr_on_plateau(values, lr0, mode='max', factor=0.5, patience=4, min_lr=1e-5, epsilon=1e-4):
    """Keras 2.1.2 ReduceLROnPlateau replayed on a monitored series.  Returns the lr
    logged at each epoch (logs['lr'] is written BEFORE the update of that epoch)."""
    best = -np.inf if mode == 'max' else np.inf
    wait, lr, logged = 0, np.float32(lr0), []
    lr_eps = min_lr * 1e-4
    for v in values:
        logged.append(float(lr))
        better = v > best + epsilon if mode == 'max' else v < best - epsilon
        if better:
            best, wait = v, 0
        else:
            if wait >= patience:
                if float(lr) > min_lr + lr_eps:
                    lr = np.float32(max(float(lr) * factor, min_lr))
                    wait = 0
            wait += 1
    return logged


def prepare_model_settings(label_count, sample_rate, clip_duration_ms, window_size_ms, window_stride_ms,
                           dct_coefficient_count, num_log_mel_features, output_representation='raw'):
    """model.py:1785-1829"""
    desired = int(sample_rate * clip_duration_ms / 1000)
    win = int(sample_rate * window_size_ms / 1000)
    stride = int(sample_rate * window_stride_ms / 1000)
    length = 0 if desired - win < 0 else 1 + int((desired - win) / stride)
    fp = {'mfcc': num_log_mel_features * length, 'raw': desired, 'spec': 257 * length,
          'mfcc_and_raw': num_log_mel_features * length}[output_representation]
    return {'desired_samples': desired, 'window_size_samples': win, 'window_stride_samples': stride,
            'spectrogram_length': length, 'spectrogram_frequencies': 257,
            'dct_coefficient_count': dct_coefficient_count, 'fingerprint_size': fp, 'label_count': label_count,
            'sample_rate': sample_rate, 'num_log_mel_features': num_log_mel_features}
