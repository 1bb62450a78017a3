#!/usr/bin/env python
"""Generates the golden fixtures under tests/golden/ FROM THE REFERENCE ITSELF.

Run in the build container only (needs /root/reference; the GPU box never sees it):
    python tests/golden/make_golden.py

What is captured (pins K1/K3/K5/K6):
  k5_control_logic.json  - the reference's own input_data.py / utils.py / classes.py / model.py
      imported with stub `tensorflow` / `keras` modules (their arithmetic ops are never executed) and
      driven with a recording fake session: SHA-1 partition of file names, data-index order,
      the NumPy-global-RNG draw order of AudioProcessor.get_data (file, time shift, background slice,
      volumes, labels) for seeded runs, data_gen's offset/epoch logic, prepare_model_settings, label maps.
  k1_graph_constants.json - variable shapes and scalar constants decoded from the graph_defs embedded
      in the reference's TensorBoard event files (logs_106 / logs_195 / logs_206).
  k3_scalars.json - the logged per-epoch scalar series (lr, val_categorical_accuracy, val_loss, ...).
Nothing of the reference's source text is stored: fixtures are inputs and observed outputs only.
"""
from __future__ import print_function

import glob
import importlib
import re
import json
import os
import struct
import sys
import tempfile
import types
from unittest import mock

import numpy as np

REF = '/root/reference'
OUT = os.path.dirname(os.path.abspath(__file__))


# ------------------------------------------------------------------------------------------------
# stub import of the reference's control logic
# ------------------------------------------------------------------------------------------------
class _StubFinder(object):
    ROOTS = ('tensorflow', 'keras', 'pandas_ml', 'IPython')

    def find_module(self, name, path=None):
        return self if name.split('.')[0] in self.ROOTS else None

    def load_module(self, name):
        if name in sys.modules:
            return sys.modules[name]
        m = mock.MagicMock(name=name)
        m.__path__ = []
        m.__all__ = []
        m.__name__ = name
        m.__loader__ = self
        m.__spec__ = None
        sys.modules[name] = m
        return m

    # importlib protocol (py3)
    def find_spec(self, name, path=None, target=None):
        if name.split('.')[0] not in self.ROOTS:
            return None
        from importlib.machinery import ModuleSpec
        return ModuleSpec(name, self, is_package=True)

    def create_module(self, spec):
        m = mock.MagicMock(name=spec.name)
        m.__path__ = []
        m.__all__ = []
        if spec.name == 'keras.layers':   # model.py:3 does `from keras.layers import *`
            m.__all__ = ['Lambda', 'Input', 'Dense', 'Dropout', 'Flatten', 'Conv1D', 'Conv2D', 'Activation',
                         'BatchNormalization', 'Multiply', 'Add', 'Concatenate', 'GlobalMaxPool1D',
                         'GlobalAveragePooling1D', 'MaxPool1D', 'Reshape', 'GRU', 'Bidirectional',
                         'MaxPooling1D', 'MaxPooling2D', 'AveragePooling1D', 'SeparableConv2D', 'Permute',
                         'GlobalAveragePooling2D', 'GlobalMaxPooling2D', 'ZeroPadding1D', 'TimeDistributed']
        return m

    def exec_module(self, module):
        pass


class _Placeholder(object):
    def __init__(self, *a, **k):
        self.name = k.get('name')


class _FakeDecoded(object):
    def __init__(self, n):
        self.audio = np.zeros((n, 1), np.float32)


class RecordingSession(object):
    """Stands in for tf.Session: records every feed_dict of sess.run and returns zeros."""

    def __init__(self, **_):
        self.calls = []

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False

    def run(self, fetches, feed_dict=None):
        self.calls.append(dict(feed_dict or {}))
        if BACKGROUND_MODE[0]:
            return _FakeDecoded(BACKGROUND_MODE[0])
        if isinstance(fetches, (list, tuple)):
            return [np.zeros((1, OUT_DIM[0]))] * len(fetches)
        return np.zeros((1, OUT_DIM[0]))


BACKGROUND_MODE = [0]
OUT_DIM = [16000]


def import_reference():
    sys.meta_path.insert(0, _StubFinder())
    sys.path.insert(0, REF)
    import tensorflow as tf
    from tensorflow.python.platform import gfile
    from tensorflow.python.util import compat
    compat.as_bytes = lambda s: s.encode('utf-8') if isinstance(s, str) else s
    gfile.Glob = lambda p: sorted(glob.glob(p))
    tf.placeholder = _Placeholder
    tf.Session = RecordingSession
    tf.Graph = lambda: None
    mods = {}
    for name in ('utils', 'input_data', 'classes', 'model'):
        sys.modules.pop(name, None)
        mods[name] = importlib.import_module(name)
    return mods


def make_tree(root, words, per_word, n_pseudo, with_noise):
    """Empty `<word>/<hash>_nohash_<k>.wav` files (content is never read by the control logic)."""
    rng = np.random.RandomState(7)
    files = []
    for w in words:
        d = os.path.join(root, w)
        os.makedirs(d)
        for i in range(per_word):
            h = '%08x' % rng.randint(0, 2 ** 31 - 1)
            fn = os.path.join(d, '%s_nohash_%d.wav' % (h, i % 3))
            open(fn, 'wb').close()
            files.append(fn)
    # pseudo-labelled clips have no _nohash_ in their name (input_data.py:94-95)
    for i in range(n_pseudo):
        w = words[i % len(words)]
        fn = os.path.join(root, w, 'clip_%05d.wav' % i)
        open(fn, 'wb').close()
        files.append(fn)
    if with_noise:
        d = os.path.join(root, '_background_noise_')
        os.makedirs(d)
        for n in ('a_noise.wav', 'b_noise.wav'):
            open(os.path.join(d, n), 'wb').close()
    return files


def capture_k5(mods):
    input_data, utils, classes, model = mods['input_data'], mods['utils'], mods['classes'], mods['model']
    out = {}
    # ---- which_set on a fixed list of names ----------------------------------------------------------
    names = ['data/train/audio/yes/%08x_nohash_%d.wav' % (i * 2654435761 % (2 ** 32), i % 4) for i in range(200)]
    names += ['data/heng_pseudo/no/clip_%d.wav' % i for i in range(5)]
    names += ['data/train/audio/unknown_unknown/%08x_nohash_0.wav' % i for i in range(5)]
    out['which_set'] = [{'name': n, 'v10_t0': input_data.which_set(n, 10.0, 0.0),
                         'v10_t10': input_data.which_set(n, 10.0, 10.0)} for n in names]
    # ---- settings / label maps ------------------------------------------------------------------------
    cases = [dict(label_count=12, sample_rate=16000, clip_duration_ms=1000, window_size_ms=30.0,
                  window_stride_ms=10.0, dct_coefficient_count=80, num_log_mel_features=60,
                  output_representation=rep) for rep in ('raw', 'mfcc', 'spec', 'mfcc_and_raw')]
    cases.append(dict(label_count=32, sample_rate=16000, clip_duration_ms=1000, window_size_ms=25.0,
                      window_stride_ms=15.0, dct_coefficient_count=80, num_log_mel_features=60,
                      output_representation='raw'))
    cases.append(dict(label_count=32, sample_rate=16000, clip_duration_ms=1000, window_size_ms=30.0,
                      window_stride_ms=10.0, dct_coefficient_count=40, num_log_mel_features=40,
                      output_representation='mfcc'))
    out['model_settings'] = [{'args': c, 'result': model.prepare_model_settings(**c)} for c in cases]
    out['classes'] = {
        'wanted': classes.get_classes(wanted_only=True),
        'all': classes.get_classes(wanted_only=False),
        'all_reversed': classes.get_classes(wanted_only=False, extend_reversed=True),
        'int2label_wanted': {str(k): v for k, v in classes.get_int2label(wanted_only=True).items()},
        'label2int_all': dict(classes.get_label2int(wanted_only=False)),
        'words_list': input_data.prepare_words_list(['yes', 'no']),
    }
    # ---- index + sampler on a synthetic directory tree ----------------------------------------------
    tmp = tempfile.mkdtemp(prefix='kws_golden_')
    cwd = os.getcwd()
    os.chdir(tmp)
    try:
        wanted = classes.get_classes(wanted_only=True)
        words = wanted + ['bed', 'cat', 'tree', 'wow']
        make_tree('data', words, per_word=40, n_pseudo=60, with_noise=True)
        settings = model.prepare_model_settings(12, 16000, 1000, 30.0, 10.0, 80, 60, 'raw')
        BACKGROUND_MODE[0] = 60 * 16000          # the two noise recordings "decode" to 60 s of zeros
        ap = input_data.AudioProcessor(['data'], 13.0, 60.0, wanted, 10.0, 0.0, settings, 'raw')
        BACKGROUND_MODE[0] = 0
        out['index'] = {
            'words': words, 'silence_percentage': 13.0, 'unknown_percentage': 60.0, 'validation_percentage': 10.0,
            'set_sizes': {k: ap.set_size(k) for k in ('training', 'validation', 'testing', 'pseudo')},
            'word_to_index': ap.word_to_index,
            'data_index': {k: [[e['label'], e['file']] for e in v] for k, v in ap.data_index.items()},
            'tree': sorted(glob.glob('data/*/*.wav')),
        }

        def record(gen, sess, n_batches):
            rec = []
            for _ in range(n_batches):
                start = len(sess.calls)
                X, y = next(gen)
                batch = []
                for fd in sess.calls[start:]:
                    bg = fd[ap.background_data_placeholder_]
                    batch.append({
                        'file': fd[ap.wav_filename_placeholder_],
                        'time_shift': int(fd[ap.time_shift_placeholder_]),
                        'bg_volume': float(fd[ap.background_volume_placeholder_]),
                        'fg_volume': float(fd[ap.foreground_volume_placeholder_]),
                        'bg_nonzero': bool(np.any(bg != 0)),
                    })
                rec.append({'feeds': batch, 'labels': np.asarray(y).argmax(axis=1).tolist(),
                            'x_shape': list(np.asarray(X).shape), 'x_dtype': str(np.asarray(X).dtype),
                            'y_dtype': str(np.asarray(y).dtype)})
            return rec
        runs = []
        for seed, kw in ((1234, dict(batch_size=16, mode='training', pseudo_frequency=0.6)),
                         (7, dict(batch_size=8, mode='training', pseudo_frequency=0.33, flip_frequency=0.5,
                                  time_shift_range=[-500, 100], background_frequency=0.8,
                                  foreground_frequency=0.9)),
                         (99, dict(batch_size=5, mode='validation', pseudo_frequency=0.0)),
                         (5, dict(batch_size=7, mode='pseudo'))):
            np.random.seed(seed)
            sess = RecordingSession()
            gen = utils.data_gen(ap, sess, **kw)
            n = 4 if kw['mode'] == 'training' else max(4, ap.set_size(kw['mode']) // kw['batch_size'] + 3)
            # the background slice start is observable only through RNG state: replay it with a
            # patched randint that logs (low, high) -> value for calls with two arguments
            log = []
            real_randint = np.random.randint

            def spy(*a, **k):
                v = real_randint(*a, **k)
                log.append([list(map(int, a)), int(v)])
                return v
            np.random.randint = spy
            try:
                rec = record(gen, sess, n)
            finally:
                np.random.randint = real_randint
            runs.append({'seed': seed, 'kwargs': kw, 'batches': rec, 'randint_log': log})
        out['sampler_runs'] = runs
    finally:
        os.chdir(cwd)
    out['tree_root_note'] = 'paths are relative to a temp dir; tests rebuild the same tree (empty files) under tmp_path'
    return out


# ------------------------------------------------------------------------------------------------
# tfevents / protobuf wire-format walkers
# ------------------------------------------------------------------------------------------------
def _varint(b, i):
    r, s = 0, 0
    while True:
        c = b[i]
        i += 1
        r |= (c & 0x7F) << s
        if not c & 0x80:
            return r, i
        s += 7


def _fields(b):
    i, n = 0, len(b)
    while i < n:
        key, i = _varint(b, i)
        f, wt = key >> 3, key & 7
        if wt == 0:
            v, i = _varint(b, i)
        elif wt == 1:
            v = b[i:i + 8]
            i += 8
        elif wt == 2:
            l, i = _varint(b, i)
            v = b[i:i + l]
            i += l
        elif wt == 5:
            v = b[i:i + 4]
            i += 4
        else:
            raise ValueError('wire type %d' % wt)
        yield f, wt, v


def _records(path):
    with open(path, 'rb') as f:
        data = f.read()
    i = 0
    while i + 12 <= len(data):
        (l,) = struct.unpack('<Q', data[i:i + 8])
        yield data[i + 12:i + 12 + l]
        i += 12 + l + 4


def _tensor(b):
    dtype, shape, content, fvals, ivals = None, [], None, [], []
    for f, wt, v in _fields(b):
        if f == 1:
            dtype = v
        elif f == 2:
            for f2, _, v2 in _fields(v):
                if f2 == 2:
                    for f3, _, v3 in _fields(v2):
                        if f3 == 1:
                            shape.append(v3)
        elif f == 4:
            content = v
        elif f == 5:
            fvals += list(struct.unpack('<%df' % (len(v) // 4), v)) if wt == 2 else [struct.unpack('<f', v)[0]]
        elif f == 7:
            if wt == 2:
                j = 0
                while j < len(v):
                    x, j = _varint(v, j)
                    ivals.append(x)
            else:
                ivals.append(v)
    if content is not None and dtype == 1:
        fvals = list(struct.unpack('<%df' % (len(content) // 4), content))
    if content is not None and dtype == 3:
        ivals = list(struct.unpack('<%di' % (len(content) // 4), content))
    return dtype, shape, fvals, ivals


def decode_events(path):
    scalars, nodes = {}, []
    for rec in _records(path):
        step, graph, summary = 0, None, None
        for f, wt, v in _fields(rec):
            if f == 2:
                step = v
            elif f == 4:
                graph = v
            elif f == 5:
                summary = v
        if graph is not None:
            for f, wt, v in _fields(graph):
                if f == 1:
                    node = {'attr': {}}
                    for f2, _, v2 in _fields(v):
                        if f2 == 1:
                            node['name'] = v2.decode()
                        elif f2 == 2:
                            node['op'] = v2.decode()
                        elif f2 == 5:
                            key, val = None, None
                            for f3, _, v3 in _fields(v2):
                                if f3 == 1:
                                    key = v3.decode()
                                elif f3 == 2:
                                    val = v3
                            node['attr'][key] = val
                    nodes.append(node)
        if summary is not None:
            for f, wt, v in _fields(summary):
                if f == 1:
                    tag, val = None, None
                    for f2, wt2, v2 in _fields(v):
                        if f2 == 1:
                            tag = v2.decode()
                        elif f2 == 2 and wt2 == 5:
                            val = struct.unpack('<f', v2)[0]
                    if tag is not None and val is not None:
                        scalars.setdefault(tag, []).append([int(step), float(val)])
    return scalars, nodes


def _attr_shape(val):
    for f, _, v in _fields(val):
        if f == 7:
            dims = []
            for f2, _, v2 in _fields(v):
                if f2 == 2:
                    for f3, _, v3 in _fields(v2):
                        if f3 == 1:
                            dims.append(v3)
            return dims
    return None


def _attr_tensor(val):
    for f, _, v in _fields(val):
        if f == 8:
            return _tensor(v)
    return None


def capture_events():
    k1, k3 = {}, {}
    for log in ('logs_106', 'logs_195', 'logs_206'):
        best = None
        for path in sorted(glob.glob(os.path.join(REF, log, 'events.out.tfevents.*'))):
            scalars, nodes = decode_events(path)
            if best is None or len(scalars.get('lr', [])) > len(best[0].get('lr', [])):
                best = (scalars, nodes, os.path.basename(path))
        scalars, nodes, fname = best
        k3[log] = {'file': fname, 'scalars': {k: v for k, v in scalars.items()}}
        variables, consts, ops = {}, {}, {}
        for n in nodes:
            ops[n.get('op')] = ops.get(n.get('op'), 0) + 1
            if n.get('op') in ('VariableV2', 'Variable') and 'shape' in n['attr']:
                if '/' in n['name'] and not n['name'].startswith('training/'):
                    variables[n['name']] = _attr_shape(n['attr']['shape'])
            if n.get('op') == 'Const' and 'value' in n['attr']:
                t = _attr_tensor(n['attr']['value'])
                if t is None:
                    continue
                dtype, shape, fvals, ivals = t
                if not shape and (len(fvals) == 1 or len(ivals) == 1):
                    consts[n['name']] = fvals[0] if fvals else ivals[0]
        keep = {}
        for name, v in consts.items():
            low = name.lower()
            if re.match(r'training/RMSprop/(Const_\d+|add_\d\d+/y)$', name):
                continue   # 100+ identical clip bounds / epsilons: one of each is kept below
            if any(s in low for s in ('stft', 'rmsprop/', 'batch_normalization_1/', 'dropout_1/', 'dct', 'rsqrt',
                                      'linspace', 'loss/', 'activation_1/', 'kernel/regularizer', 'add/y',
                                      'conv1d_1/')):
                keep[name] = v
        k1[log] = {'file': fname, 'variables': variables, 'constants': keep, 'op_census': ops,
                   'n_nodes': len(nodes)}
    return k1, k3


def main():
    if not os.path.isdir(REF):
        raise SystemExit('reference checkout not found at %s' % REF)
    mods = import_reference()
    k5 = capture_k5(mods)
    with open(os.path.join(OUT, 'k5_control_logic.json'), 'w') as f:
        json.dump(k5, f, indent=0, sort_keys=True)
    k1, k3 = capture_events()
    with open(os.path.join(OUT, 'k1_graph_constants.json'), 'w') as f:
        json.dump(k1, f, indent=0, sort_keys=True)
    with open(os.path.join(OUT, 'k3_scalars.json'), 'w') as f:
        json.dump(k3, f, indent=0, sort_keys=True)
    print('wrote fixtures:', [(p, os.path.getsize(os.path.join(OUT, p))) for p in
                              ('k5_control_logic.json', 'k1_graph_constants.json', 'k3_scalars.json')])


if __name__ == '__main__':
    main()
