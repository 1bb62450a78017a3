"""Value-level checks of the product's HOST-side pieces that round 1 only exercised end to end:

  * speech_recognition_amd.features tables (what AudioProcessor / AudioConverter upload into the STFT plan) against
    the oracle's tables (input_data.py:361-381, audio.py:15-23);
  * ConfusionMatrixCallback / log_loss (row a16) against outputs of the reference's own callbacks.py (fixture K8,
    tests/golden/make_golden_callback.py) and against hand-computed values;
  * WAV ingest (row f1): _read_wav_int16 / load_wav_file / save_wav_file against scipy.io.wavfile on mono, stereo,
    odd-sized LIST chunks, short and long files (input_data.py:117-156, 335-336);
  * tta.shard_range (config C5's multi-GPU split): disjoint cover;
  * the oracle's layer lengths follow Keras' 'same' pooling / strided convolution (ceil), odd lengths included.
No GPU, no libkws_hip.so compute call."""
import json
import os
import struct

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


# ---- feature tables ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("win,n_mel,n_keep", [(480, 80, 60), (480, 40, 40), (400, 80, 60), (400, 40, 40)])
def test_path_b_tables_match_oracle(win, n_mel, n_keep):
    from oracle import features as OF
    from speech_recognition_amd.features import path_b_tables
    t = path_b_tables(win, n_mel, n_keep, 16000)
    o = OF.tables_path_b(win, n_mel, n_keep)
    assert t['fft_length'] == o['fft_length'] == 512
    assert t['window'].dtype == np.float32 and t['mel'].dtype == np.float32 and t['dct'].dtype == np.float32
    assert t['mel'].shape == (257, n_mel) and t['dct'].shape == (n_mel, n_keep) and t['window'].shape == (win,)
    # tf.contrib.signal computes these tables in float32 in-graph: identical arithmetic -> identical bits
    np.testing.assert_array_equal(t['window'], np.asarray(o['window'], np.float32))
    np.testing.assert_array_equal(t['mel'], np.asarray(o['mel'], np.float32))
    np.testing.assert_allclose(t['dct'], np.asarray(o['dct'], np.float32), rtol=0, atol=1e-7)
    assert t['log_offset'] == o['log_offset'] == 1e-6 and t['log_floor'] == o['log_floor'] == 0.0
    assert (t['mel'][0] == 0).all()                    # DC bin excluded (linear_to_mel_weight_matrix)


def test_path_a_tables_match_oracle():
    from oracle import features as OF
    from speech_recognition_amd.features import path_a_tables
    t = path_a_tables(480, 16000, 40, 40)
    o = OF.tables_path_a(480, 16000, 40, 40)
    assert t['fft_length'] == o['fft_length'] == 512
    np.testing.assert_allclose(t['window'], o['window'], rtol=0, atol=6e-8)        # double table rounded to f32
    np.testing.assert_allclose(t['mel'], o['mel'], rtol=0, atol=6e-8)
    np.testing.assert_allclose(t['dct'], o['dct'], rtol=0, atol=6e-8)
    assert (t['mel'] != 0).sum() == (np.asarray(o['mel']) != 0).sum()               # same filter support
    assert t['log_offset'] == 0.0 and t['log_floor'] == o['log_floor'] == 1e-12


# ---- a16: validation callback --------------------------------------------------------------------------------
class _FakeModel(object):
    def __init__(self, preds):
        self.preds = list(preds)

    def predict(self, X):
        return self.preds.pop(0)


def _run_callback(tmp_path, words, wanted, y_true_idx, y_pred, epoch=7):
    from speech_recognition_amd.callbacks import ConfusionMatrixCallback
    C = len(words)
    cwd = os.getcwd()
    os.chdir(str(tmp_path))
    try:
        ys = [np.eye(C, dtype=np.float32)[np.asarray(b)] for b in y_true_idx]
        ps = [np.asarray(p, dtype=np.float32) for p in y_pred]
        gen = iter([(np.zeros((len(y), 4)), y) for y in ys])
        cb = ConfusionMatrixCallback(gen, len(ys), wanted, words, {w: i for i, w in enumerate(words)})
        cb.set_model(_FakeModel(ps))
        logs = {}
        cb.on_epoch_end(epoch, logs)
        with open('confusion_matrix.txt') as f:
            text = f.read()
        with open('wanted_confusion_matrix.txt') as f:
            wtext = f.read()
    finally:
        os.chdir(cwd)
    return logs, text, wtext


def test_confusion_matrix_callback_matches_reference_outputs(tmp_path):
    with open(os.path.join(HERE, 'golden', 'k8_callback.json')) as f:
        gold = json.load(f)
    from speech_recognition_amd.callbacks import log_loss
    for i, c in enumerate(gold['cases']):
        d = tmp_path / ("c%d" % i)
        d.mkdir()
        logs, text, wtext = _run_callback(d, c['words'], c['wanted'], c['y_true'], c['y_pred'])
        assert set(logs) == set(c['logs'])
        for k, v in c['logs'].items():
            assert abs(float(logs[k]) - v) <= 1e-7 * max(1.0, abs(v)), (c['name'], k, logs[k], v)
        assert text.split('\n')[:2] == c['acc_line'] and wtext.split('\n')[:2] == c['acc_line']
        C = len(c['words'])
        yt = np.concatenate([np.eye(C, dtype=np.float32)[np.asarray(b)] for b in c['y_true']])
        yp = np.concatenate([np.asarray(p, np.float32) for p in c['y_pred']])
        assert abs(log_loss(yt, yp) - c['log_loss']) <= 1e-7 * abs(c['log_loss'])


def test_confusion_matrix_callback_hand_computed(tmp_path):
    """4 classes, 8 clips; class 'c' never occurs as truth but is predicted once (an empty row: accuracy 0.0, the
    reference's `if num:` branch), 'b' is unwanted and folds into _unknown_."""
    words = ['_silence_', '_unknown_', 'a', 'b', 'c']
    wanted = ['a', 'c']
    y_true = [[0, 0, 1, 2], [2, 2, 3, 3]]
    pred_idx = [[0, 1, 1, 2], [2, 4, 3, 1]]
    # probabilities: 0.7 on the predicted class, 0.075 elsewhere
    y_pred = [[[0.7 if j == p else 0.075 for j in range(5)] for p in b] for b in pred_idx]
    logs, text, wtext = _run_callback(tmp_path, words, wanted, y_true, y_pred, epoch=3)
    # correct: clip0 (0->0), clip2 (1->1), clip3 (2->2), clip4 (2->2), clip6 (3->3) = 5/8
    assert logs['val_categorical_accuracy'] == 5.0 / 8.0
    # rows (actual) sorted by label string: _silence_ 1/2, _unknown_ 1/1, a 2/3, b 1/2, c 0/0 -> 0.0
    per_class = np.float32([0.5, 1.0, 2.0 / 3.0, 0.5, 0.0])
    assert logs['val_mean_categorical_accuracy_all'] == per_class.mean()
    # wanted view: every label outside `wanted` folds into _unknown_ - here also _silence_, which train.py keeps by
    # passing prepare_words_list(...) - rows: _unknown_ (clips 0,1,2,6,7, all predicted outside `wanted`) 5/5, a 2/3, c 0/0
    wanted_acc = np.float32([1.0, 2.0 / 3.0, 0.0])
    assert logs['val_mean_categorical_accuracy_wanted'] == wanted_acc.mean()
    # log loss: the true class got 0.7 in 5 clips and 0.075 in 3 (float32 probabilities, as np.float32(y_pred))
    expect = -(5 * np.log(np.float32(0.7)) + 3 * np.log(np.float32(0.075))) / 8.0
    assert abs(logs['val_loss'] - expect) < 1e-6
    assert text.split('\n')[1] == "[003]: val_categorical_accuracy: 0.62, val_mean_categorical_accuracy_wanted: %.2f" % wanted_acc.mean()
    # matrix layout: rows = actual, columns = predicted
    from speech_recognition_amd.callbacks import confusion_matrix
    t = [words[i] for b in y_true for i in b]
    p = [words[i] for b in pred_idx for i in b]
    labels, m = confusion_matrix(t, p)
    assert labels == ['_silence_', '_unknown_', 'a', 'b', 'c']
    assert m.tolist() == [[1, 1, 0, 0, 0], [0, 1, 0, 0, 0], [0, 0, 2, 0, 1], [0, 1, 0, 1, 0], [0, 0, 0, 0, 0]]


# ---- f1: WAV ingest ------------------------------------------------------------------------------------------
def _riff(chunks):
    body = b'WAVE' + b''.join(cid + struct.pack('<I', len(b)) + b + (b'\x00' if len(b) & 1 else b'') for cid, b in chunks)
    return b'RIFF' + struct.pack('<I', len(body)) + body


def test_wav_reader_against_scipy(tmp_path):
    from scipy.io import wavfile
    from speech_recognition_amd.input_data import _read_wav_int16, load_wav_file, save_wav_file
    rng = np.random.RandomState(0)
    mono = rng.randint(-32768, 32768, 16000).astype(np.int16)
    mono[:3] = [-32768, 32767, 0]
    short = rng.randint(-32768, 32768, 9000).astype(np.int16)
    long_ = rng.randint(-32768, 32768, 20000).astype(np.int16)
    stereo = rng.randint(-32768, 32768, (5000, 2)).astype(np.int16)
    for name, arr in (('mono', mono), ('short', short), ('long', long_), ('stereo', stereo)):
        fn = str(tmp_path / (name + '.wav'))
        wavfile.write(fn, 16000, arr)
        rate0, ref = wavfile.read(fn)
        a, rate = _read_wav_int16(fn)
        assert rate == rate0 == 16000 and a.dtype == np.int16
        first = ref if ref.ndim == 1 else ref[:, 0]         # desired_channels=1: the first channel (input_data.py:335)
        np.testing.assert_array_equal(a, first)
        f = load_wav_file(fn)
        assert f.dtype == np.float32
        np.testing.assert_array_equal(f, first.astype(np.float32) / np.float32(32768.0))   # DecodeWav scale
    # a LIST chunk of ODD size between fmt and data (padded to even, as ffmpeg / sox write it), and trailing junk
    fmt = struct.pack('<HHIIHH', 1, 1, 16000, 32000, 2, 16)
    pcm = mono[:1001].astype('<i2').tobytes()
    fn = str(tmp_path / 'list.wav')
    with open(fn, 'wb') as f:
        f.write(_riff([(b'fmt ', fmt), (b'LIST', b'INFOISFT\x05\x00\x00\x00Lavf\x00'[:21]), (b'data', pcm), (b'junk', b'xyz')]))
    a, rate = _read_wav_int16(fn)
    np.testing.assert_array_equal(a, mono[:1001])
    _, ref = wavfile.read(fn)
    np.testing.assert_array_equal(a, ref)
    # errors: not RIFF, 8-bit PCM
    bad = str(tmp_path / 'bad.wav')
    with open(bad, 'wb') as f:
        f.write(b'RIFX' + b'\x00' * 40)
    with pytest.raises(ValueError):
        _read_wav_int16(bad)
    with open(bad, 'wb') as f:
        f.write(_riff([(b'fmt ', struct.pack('<HHIIHH', 1, 1, 16000, 16000, 1, 8)), (b'data', b'\x80' * 10)]))
    with pytest.raises(ValueError):
        _read_wav_int16(bad)
    # EncodeWav round trip (input_data.py:135-156): clamp, x 32767, truncate
    x = np.array([0.0, 0.5, -0.5, 1.0, -1.0, 1.5, -1.5, 1e-5], np.float32)
    out = str(tmp_path / 'enc.wav')
    save_wav_file(out, x.reshape(-1, 1), 16000)
    rate, back = wavfile.read(out)
    assert rate == 16000
    np.testing.assert_array_equal(back, (np.clip(x, -1, 1) * 32767.0).astype(np.int16))


def test_bank_rows_pad_or_crop(tmp_path):
    """The int16 clip bank row of a file = its first desired_samples samples, zero padded (DecodeWav desired_samples,
    input_data.py:335-336) - exercised through the same helper AudioProcessor._build_bank uses, without a GPU."""
    from scipy.io import wavfile
    from speech_recognition_amd.input_data import bank_from_files
    rng = np.random.RandomState(1)
    files = {}
    for i, n in enumerate((16000, 9000, 20000, 1)):
        fn = str(tmp_path / ('f%d.wav' % i))
        wavfile.write(fn, 16000, rng.randint(-32768, 32768, n).astype(np.int16))
        files[fn] = i
    bank = bank_from_files(files, 16000)
    assert bank.shape == (4, 16000) and bank.dtype == np.int16
    for fn, r in files.items():
        _, ref = wavfile.read(fn)
        n = min(len(ref), 16000)
        np.testing.assert_array_equal(bank[r, :n], ref[:n])
        assert (bank[r, n:] == 0).all()


# ---- C5 split ----------------------------------------------------------------------------------------------------
def test_tta_shard_range_is_a_disjoint_cover(monkeypatch):
    from speech_recognition_amd import parallel, tta
    for n in (0, 1, 7, 158538):
        for W in (1, 2, 8):
            got = []
            for r in range(W):
                monkeypatch.setattr(parallel, 'world_size', lambda W=W: W)
                monkeypatch.setattr(parallel, 'rank', lambda r=r: r)
                lo, hi = tta.shard_range(n)
                assert 0 <= lo <= hi <= n
                got.append((lo, hi))
            assert got[0][0] == 0 and got[-1][1] == n
            for (a, b), (c, d) in zip(got[:-1], got[1:]):
                assert b == c                                           # contiguous, no gap, no overlap
            sizes = [b - a for a, b in got]
            assert max(sizes) - min([s for s in sizes] or [0]) <= max(1, (n + W - 1) // W) and sum(sizes) == n
            assert max(sizes) <= (n + W - 1) // W                        # no rank carries more than ceil(n/W)


# ---- oracle refuses what the device refuses ----------------------------------------------------------------------
def test_logmfcc_oracle_lengths_follow_keras_same_layers():
    """MaxPool1D(2, 2, 'same') and Conv1D(nf, 1, strides=2, 'same') both give ceil(L / 2) (model.py:1431-1441): the
    reference function's default spectrogram_length = 65 runs 63 -> 32 -> 16 -> 8."""
    from oracle.net import LogMfccNet
    net = LogMfccNet(num_classes=32, spectrogram_length=65, num_features=40)
    assert [b['Lout'] for b in net.blocks] == [63, 63, 32, 32, 16, 16, 16, 8, 8, 8] and net.T == 8
    net = LogMfccNet(num_classes=32, spectrogram_length=98, num_features=40)
    assert [b['Lout'] for b in net.blocks] == [96, 96, 48, 48, 24, 24, 24, 12, 12, 12]
    with pytest.raises(ValueError):
        LogMfccNet(num_classes=32, spectrogram_length=2, num_features=40)


def test_sampler_refuses_background_no_longer_than_a_clip():
    """np.random.randint(0, len(bg) - desired_samples) raises ValueError when a noise recording is <= 1 s
    (input_data.py:485); the native sampler must do the same, not index outside the recording."""
    from speech_recognition_amd.sampler import DataIndex
    idx = DataIndex.from_entries({'training': [(0, 'yes'), (1, 'no')], 'validation': [], 'testing': [], 'pseudo': []},
                                 ['yes', 'no'])
    args = dict(mode='training', offset=0, sample_count=4, how_many=4, desired_samples=16000,
                background_lengths=[16000], background_starts=[0], background_frequency=0.5, background_volume_range=0.1,
                foreground_frequency=0.5, foreground_volume_range=0.1, time_shift_frequency=0.5, time_shift_range=[-100, 0],
                pseudo_frequency=0.0, flip_frequency=0.0, silence_volume_range=0.0)
    np.random.seed(3)
    with pytest.raises(ValueError):
        idx.draw_python(**args)
    np.random.seed(3)
    with pytest.raises(ValueError):
        idx.draw(**args)
    args['background_lengths'] = [16001]
    np.random.seed(3)
    a = idx.draw_python(**args)
    np.random.seed(3)
    b = idx.draw(**args)
    for u, v in zip(a, b):
        np.testing.assert_array_equal(u, v)
    assert (b[3] == 0).all()            # the only legal offset
