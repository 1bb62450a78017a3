"""Export side of the pseudo-label loop: the offline twins of the reference's
convert_from_see_v3_bugfix.py and create_pseudo_with_thresh.py as functions, so that train -> predict (TTA) ->
relabel runs around the accelerated path without the hard-coded file names and row counts of the scripts.
Pure host-side NumPy / file work; pinned bit for bit by tests/golden/k7_export_tools.json, which the reference's
own scripts produced (tests/golden/make_golden_export.py)."""
import os
import shutil

import numpy as np

SILENCE_LABEL = '_silence_'
AUDIO_NAMES = ['silence', 'unknown', 'yes', 'no', 'up', 'down', 'left', 'right', 'on', 'off', 'stop', 'go']


def head32to12_offline(all_probs, int2label, audio_names=AUDIO_NAMES):
    """convert_from_see_v3_bugfix.py:76-100: the 12 submission classes from the 32-class probabilities - wanted
    words by name, silence from column 0, 'unknown' = max over every other word, then a softmax over the 12
    PROBABILITIES (float32 throughout, like the script).  `kws_head32to12` is the device form of the same map."""
    all_probs = np.asarray(all_probs)
    out = np.zeros((all_probs.shape[0], len(audio_names)), np.float32)
    unknown = []
    for i, name in int2label.items():
        if name == SILENCE_LABEL:
            continue
        if name in audio_names:
            out[:, audio_names.index(name)] = all_probs[:, i]
        else:
            unknown.append(all_probs[:, i])
    out[:, 0] = all_probs[:, 0]
    out[:, 1] = np.float32(unknown).max(axis=0)
    e = np.exp(out)
    return e / e.sum(axis=1, keepdims=True)


def write_probs_uint8_memmap(path, probs):
    """convert_from_see_v3_bugfix.py:107-110: probabilities * 255 truncated to uint8, one row per test clip."""
    probs = np.asarray(probs)
    mm = np.memmap(path, dtype='uint8', mode='w+', shape=probs.shape)
    mm[...] = (probs * 255)
    mm.flush()
    return mm


def make_pseudo_labels(fnames, probs_uint8, src_dir, pseudo_dir, prob_thresh=0.7, audio_names=AUDIO_NAMES,
                       silence_group=30, silence_gain=0.35):
    """create_pseudo_with_thresh.py:14-63.  Test clips whose top probability (uint8 / 255) reaches
    `prob_thresh` are copied into `pseudo_dir/<label>/`; confident 'silence' clips are concatenated in groups of
    `silence_group`, divided by `silence_gain` ("make it louder") and written as
    `_background_noise_/custom_silence_%06d.wav` (a trailing incomplete group is dropped).  A label directory is
    created as soon as a clip is PREDICTED as that label, confident or not, and an existing `pseudo_dir` is
    wiped first - both like the script.  Returns (num_labels, num_small_prob)."""
    from scipy.io import wavfile as wf
    probs_uint8 = np.asarray(probs_uint8)
    max_probs = np.float32(probs_uint8.max(axis=-1)) / 255
    preds = probs_uint8.argmax(axis=-1)
    if os.path.exists(pseudo_dir):
        shutil.rmtree(pseudo_dir)
    os.makedirs(pseudo_dir)
    num_small_prob = num_labels = silence_count = 0
    silence_data = []
    made = set()
    for i in range(len(fnames)):
        fn = fnames[i]
        label = audio_names[preds[i]]
        dir_name = os.path.join(pseudo_dir, '_background_noise_' if label == 'silence' else label)
        if dir_name not in made:
            os.makedirs(dir_name, exist_ok=True)
            made.add(dir_name)
        if max_probs[i] < prob_thresh:
            num_small_prob += 1
            continue
        src_fn = os.path.join(src_dir, fn)
        if label == 'silence':
            rate, data = wf.read(src_fn)
            silence_data.append(np.float32(data) / 32767)
            silence_count += 1
            if silence_count % silence_group == 0:
                dst_fn = os.path.join(dir_name, 'custom_silence_%06d.wav' % (silence_count // silence_group))
                wf.write(dst_fn, rate, np.int16((np.concatenate(silence_data) / silence_gain) * 32767))
                num_labels += 1
                silence_data = []
        else:
            shutil.copy(src_fn, os.path.join(dir_name, fn))
            num_labels += 1
    return num_labels, num_small_prob
