"""Host-only part of the data path: dataset index and per-batch sampler.

No GPU, no torch: this is the control logic the reference runs in Python (input_data.py:61-114,
182-272, 428-514) and it stays Python/NumPy here; the device work it parameterises is one
`kws_augment_*` launch per batch (see input_data.AudioProcessor).  The draw order of the NumPy global
RNG is the reference's and is pinned by tests/golden/k5_control_logic.json.
"""
from __future__ import absolute_import, division, print_function

import glob
import hashlib
import math
import os.path
import random
import re
import threading

import numpy as np

MAX_NUM_WAVS_PER_CLASS = 2 ** 27 - 1  # ~134M
SILENCE_LABEL = '_silence_'
SILENCE_INDEX = 0
UNKNOWN_WORD_LABEL = '_unknown_'
UNKNOWN_WORD_INDEX = 1
BACKGROUND_NOISE_DIR_NAME = '_background_noise_'
RANDOM_SEED = 59185
PARTITIONS = ('validation', 'testing', 'training', 'pseudo')


def prepare_words_list(wanted_words):
    """reference input_data.py:49-58"""
    return [SILENCE_LABEL, UNKNOWN_WORD_LABEL] + wanted_words


def which_set(filename, validation_percentage, testing_percentage):
    """Stable SHA-1 partition of a file name (reference input_data.py:61-114): directory
    `unknown_unknown` -> training; no `_nohash_` in the name -> 'pseudo'; else the hash of the part
    before `_nohash_` mapped to [0,100] decides validation / testing / training."""
    if os.path.basename(os.path.dirname(filename)) == 'unknown_unknown':
        return 'training'
    base_name = os.path.basename(filename)
    if base_name.find('_nohash_') == -1:
        return 'pseudo'
    hash_name = re.sub(r'_nohash_.*$', '', base_name)
    digest = hashlib.sha1(hash_name.encode('utf-8')).hexdigest()
    percentage_hash = ((int(digest, 16) % (MAX_NUM_WAVS_PER_CLASS + 1)) * (100.0 / MAX_NUM_WAVS_PER_CLASS))
    if percentage_hash < validation_percentage:
        return 'validation'
    if percentage_hash < (testing_percentage + validation_percentage):
        return 'testing'
    return 'training'


_RNG_LOCK = threading.Lock()     # serialises every read-modify-write of np.random's global state (DataIndex.draw)


class DataIndex(object):
    """data_index / word_to_index / words_list of the reference's AudioProcessor plus flat arrays
    (bank row, label index, is-silence) per partition for the sampler."""

    def __init__(self):
        self.data_index = {s: [] for s in PARTITIONS}
        self.word_to_index = {}
        self.words_list = []
        self.file_row = {}
        self.rows, self.labels, self.silence = {}, {}, {}

    @classmethod
    def from_dirs(cls, data_dirs, silence_percentage, unknown_percentage, wanted_words, validation_percentage,
                  testing_percentage):
        """reference input_data.py:182-272 (same ordering: random.seed(59185), sorted glob, shuffles)."""
        self = cls()
        random.seed(RANDOM_SEED)
        wanted_words_index = {w: i + 2 for i, w in enumerate(wanted_words)}
        unknown_index = {s: [] for s in PARTITIONS}
        all_words = {}
        for data_dir in data_dirs:
            search_path = os.path.join(data_dir, '*', '*.wav')
            for wav_path in sorted(glob.glob(search_path)):
                word = re.search('.*/([^/]+)/.*.wav', wav_path).group(1).lower()
                if word == BACKGROUND_NOISE_DIR_NAME:
                    continue
                all_words[word] = True
                set_index = which_set(wav_path, validation_percentage, testing_percentage)
                entry = {'label': word, 'file': wav_path}
                if word in wanted_words_index:
                    self.data_index[set_index].append(entry)
                else:
                    unknown_index[set_index].append(entry)
            if not all_words:
                raise Exception('No .wavs found at ' + search_path)
            for wanted_word in wanted_words:
                if wanted_word not in all_words:
                    raise Exception('Expected to find ' + wanted_word + ' in labels but only found ' +
                                    ', '.join(all_words.keys()))
        silence_wav_path = self.data_index['training'][0]['file']
        for set_index in PARTITIONS:
            set_size = len(self.data_index[set_index])
            silence_size = int(math.ceil(set_size * silence_percentage / 100))
            for _ in range(silence_size):
                self.data_index[set_index].append({'label': SILENCE_LABEL, 'file': silence_wav_path})
            random.shuffle(unknown_index[set_index])
            unknown_size = int(math.ceil(set_size * unknown_percentage / 100))
            self.data_index[set_index].extend(unknown_index[set_index][:unknown_size])
        for set_index in PARTITIONS:
            random.shuffle(self.data_index[set_index])
        self.words_list = prepare_words_list(wanted_words)
        for word in all_words:
            self.word_to_index[word] = wanted_words_index.get(word, UNKNOWN_WORD_INDEX)
        self.word_to_index[SILENCE_LABEL] = SILENCE_INDEX
        rows = {}
        for part in PARTITIONS:
            for e in self.data_index[part]:
                if e['file'] not in rows:
                    rows[e['file']] = len(rows)
        self.file_row = rows
        self._finish()
        return self

    @classmethod
    def from_entries(cls, entries, wanted_words):
        """Synthetic source: entries = {partition: [(bank_row, word), ...]} (bench.py / tests)."""
        self = cls()
        self.words_list = prepare_words_list(wanted_words)
        wanted_words_index = {w: i + 2 for i, w in enumerate(wanted_words)}
        self.word_to_index = {SILENCE_LABEL: SILENCE_INDEX}
        for s, lst in entries.items():
            for row, word in lst:
                fn = 'synthetic://%d' % row
                self.file_row[fn] = row
                self.data_index[s].append({'label': word, 'file': fn})
                if word != SILENCE_LABEL:
                    self.word_to_index[word] = wanted_words_index.get(word, UNKNOWN_WORD_INDEX)
        self._finish()
        return self

    def _finish(self):
        for s, p in self.data_index.items():
            self.rows[s] = np.array([self.file_row[e['file']] for e in p], dtype=np.int32)
            self.labels[s] = np.array([self.word_to_index[e['label']] for e in p], dtype=np.int32)
            self.silence[s] = np.array([e['label'] == SILENCE_LABEL for e in p], dtype=bool)

    def set_size(self, mode):
        return len(self.data_index[mode])

    def draw(self, mode, offset, sample_count, how_many, desired_samples, background_lengths, background_starts,
             background_frequency, background_volume_range, foreground_frequency, foreground_volume_range,
             time_shift_frequency, time_shift_range, pseudo_frequency, flip_frequency, silence_volume_range):
        """Native form of `draw_python` (csrc/sampler.cpp `kws_sampler_draw`): same draws from the same
        NumPy global MT19937 stream - the state is read with np.random.get_state(), advanced in C and
        written back - roughly 100x faster than the interpreter loop."""
        import ctypes
        from . import _lib
        lib = _lib.load()
        with _RNG_LOCK:
            return self._draw_locked(lib, ctypes, _lib, mode, offset, sample_count, how_many, desired_samples,
                                     background_lengths, background_starts, background_frequency, background_volume_range,
                                     foreground_frequency, foreground_volume_range, time_shift_frequency, time_shift_range,
                                     pseudo_frequency, flip_frequency, silence_volume_range)

    def _draw_locked(self, lib, ctypes, _lib, mode, offset, sample_count, how_many, desired_samples, background_lengths,
                     background_starts, background_frequency, background_volume_range, foreground_frequency,
                     foreground_volume_range, time_shift_frequency, time_shift_range, pseudo_frequency, flip_frequency,
                     silence_volume_range):
        # get_state -> C (GIL released) -> set_state is a read-modify-write of the GLOBAL generator: fit_generator's
        # enqueuer thread draws training batches while the validation callback draws on the main thread, and an
        # interleaving would replay a stretch of the stream.  The reference consumes the generator call by call under
        # the GIL and never rewinds; _RNG_LOCK gives the same guarantee here.
        st = np.random.get_state()
        if st[0] != 'MT19937':
            raise _lib.KwsError("np.random global state is not MT19937")
        key = np.ascontiguousarray(st[1], dtype=np.uint32).copy()
        pos = ctypes.c_int(int(st[2]))

        def pset(part):
            sil = np.ascontiguousarray(self.silence[part], dtype=np.uint8)
            s = _lib.SamplerSet(self.rows[part].ctypes.data, self.labels[part].ctypes.data, sil.ctypes.data,
                                len(self.rows[part]))
            return s, sil
        cand, keep1 = pset(mode)
        pseudo, keep2 = pset('pseudo')
        bg_len = np.ascontiguousarray(background_lengths, dtype=np.int64)
        bg_start = np.ascontiguousarray(background_starts, dtype=np.int64)
        a = _lib.SamplerArgs()
        a.deterministic = int(how_many == -1 or mode != 'training')
        a.offset, a.count = int(offset), int(sample_count)
        a.use_background = int(len(bg_len) > 0 and mode == 'training')
        a.n_bg = len(bg_len)
        a.bg_len, a.bg_start = bg_len.ctypes.data, bg_start.ctypes.data
        a.desired_samples = int(desired_samples)
        a.shift_lo, a.shift_hi = int(time_shift_range[0]), int(time_shift_range[1])
        a.background_frequency, a.background_volume_range = background_frequency, background_volume_range
        a.foreground_frequency, a.foreground_volume_range = foreground_frequency, foreground_volume_range
        a.time_shift_frequency, a.pseudo_frequency = time_shift_frequency, pseudo_frequency
        a.flip_frequency, a.silence_volume_range = flip_frequency, silence_volume_range
        rows = np.empty(sample_count, np.int32)
        labels = np.empty(sample_count, np.int32)
        shift = np.empty(sample_count, np.int32)
        bg_off = np.empty(sample_count, np.int64)
        bg_vol = np.empty(sample_count, np.float32)
        fg_vol = np.empty(sample_count, np.float32)
        rc = lib.kws_sampler_draw(key.ctypes.data, ctypes.byref(pos), ctypes.byref(cand), ctypes.byref(pseudo),
                                  ctypes.byref(a), rows.ctypes.data, labels.ctypes.data, shift.ctypes.data,
                                  bg_off.ctypes.data, bg_vol.ctypes.data, fg_vol.ctypes.data)
        np.random.set_state((st[0], key, pos.value, st[3], st[4]))     # draws made before an error stay consumed
        if rc != 0:
            msg = (lib.kws_last_error() or b"").decode()
            if "low >= high" in msg:          # what np.random.randint raises at input_data.py:474,485
                raise ValueError(msg)
            _lib.check(rc, "kws_sampler_draw")
        return rows, labels, shift, bg_off, bg_vol, fg_vol

    def draw_python(self, mode, offset, sample_count, how_many, desired_samples, background_lengths,
                    background_starts, background_frequency, background_volume_range, foreground_frequency,
                    foreground_volume_range, time_shift_frequency, time_shift_range, pseudo_frequency,
                    flip_frequency, silence_volume_range):
        """Per-clip augmentation parameters for one batch, drawn from the NumPy GLOBAL RNG in the
        reference's order (input_data.py:457-514): sample pick, time shift, background (recording,
        offset, volume incl. the silence special case), foreground volume / sign flip.
        Returns (rows, labels, shift, bg_off [absolute sample in the concatenated noise], bg_vol, fg_vol)."""
        rows_m, lab_m, sil_m = self.rows[mode], self.labels[mode], self.silence[mode]
        rows_p, lab_p, sil_p = self.rows['pseudo'], self.labels['pseudo'], self.silence['pseudo']
        n_cand, n_pseudo = len(rows_m), len(rows_p)
        use_background = len(background_lengths) > 0 and (mode == 'training')
        pick_deterministically = (mode != 'training')
        rows = np.empty(sample_count, np.int32)
        labels = np.empty(sample_count, np.int32)
        shift = np.zeros(sample_count, np.int32)
        bg_off = np.zeros(sample_count, np.int64)
        bg_vol = np.zeros(sample_count, np.float32)
        fg_vol = np.empty(sample_count, np.float32)
        uniform, randint = np.random.uniform, np.random.randint
        n_bg = len(background_lengths)
        for k in range(sample_count):
            i = offset + k
            if how_many == -1 or pick_deterministically:
                r, lab, sil = rows_m[i], lab_m[i], sil_m[i]
            elif uniform(0, 1) < pseudo_frequency:
                j = randint(n_pseudo)
                r, lab, sil = rows_p[j], lab_p[j], sil_p[j]
            else:
                j = randint(n_cand)
                r, lab, sil = rows_m[j], lab_m[j], sil_m[j]
            if uniform(0.0, 1.0) < time_shift_frequency:
                shift[k] = randint(time_shift_range[0], time_shift_range[1] + 1)
            if use_background:
                bi = randint(n_bg)
                bo = randint(0, background_lengths[bi] - desired_samples)
                bg_off[k] = background_starts[bi] + bo
                if uniform(0, 1) < background_frequency:
                    bg_vol[k] = uniform(0, background_volume_range)
                elif sil and uniform(0, 1) < 0.9:
                    bg_vol[k] = uniform(0, silence_volume_range)
            if sil:
                fg = 0.0
            else:
                fg = 1.0
                if uniform(0, 1) < foreground_frequency:
                    fg = 1.0 + uniform(-foreground_volume_range, foreground_volume_range)
                if uniform(0, 1) < flip_frequency:
                    fg *= -1.0
            fg_vol[k] = fg
            rows[k], labels[k] = r, lab
        return rows, labels, shift, bg_off, bg_vol, fg_vol
