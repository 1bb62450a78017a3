"""End-to-end drop-in test on the GPU.  A training caller reaches the hot path only through the module names and call
signatures the reference's scripts bind (train.py:2-11, 22-75: Session -> settings -> AudioProcessor on wav
directories incl. a pseudo-label directory -> two data_gen generators -> speech_model -> ConfusionMatrixCallback +
ReduceLROnPlateau + TensorBoard + ModelCheckpoint -> fit_generator -> evaluate_generator), run by
`python -m speech_recognition_amd.run_script` on a small generated wav dataset; a prediction caller then loads the saved
checkpoint, fetches clips through the processing graph's feed keys and forms the 3-term TTA average
(make_submission.py:86-146).  Both callers are this repository's own code (see the note above them); the values the
validation callback logs are checked against an independent recomputation."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

# The two callers below were written for this test: they reach the drop-in modules through the API surface the
# reference's scripts use (module names, call signatures, callback classes, feed keys), not through the reference's
# text.  That every name the real train.py / make_submission.py import resolves through dropin/ is checked against the
# real files in tests/test_dropin_cpu.py (build container only).
TRAIN_CALLER = r'''
import json
import sys

import numpy
import tensorflow
import keras.backend
import keras.callbacks as kc
import IPython            # noqa: F401  (the reference's train.py imports it; it must resolve)

import callbacks as validation_callbacks
import classes as vocabulary
import input_data
import model as models
import utils

BATCH = 16
EPOCHS = 3


def open_session():
    options = tensorflow.GPUOptions(per_process_gpu_memory_fraction=0.5)
    session = tensorflow.Session(config=tensorflow.ConfigProto(gpu_options=options))
    keras.backend.set_session(session)
    return session


def make_processor(words):
    labels = input_data.prepare_words_list(words)
    cfg = models.prepare_model_settings(len(labels), 16000, 1000, 30.0, 10.0, 80, 60, output_representation='raw')
    proc = input_data.AudioProcessor(['data/train/audio', 'data/heng_pseudo'], 13.0, 60.0, words, 10.0, 0.0, cfg,
                                     output_representation='raw')
    return proc, cfg, labels


def recompute_validation(net, proc, session, steps):
    # an independent pass over the validation partition with a fresh generator: what the callback must have logged
    gen = utils.data_gen(proc, session, batch_size=BATCH, mode='validation', pseudo_frequency=0.0)
    truth, pred = [], []
    for _ in range(steps):
        X, y = next(gen)
        pred.append(numpy.asarray(net.predict(X), dtype=numpy.float32))
        truth.append(numpy.asarray(y, dtype=numpy.float32))
    truth, pred = numpy.concatenate(truth), numpy.concatenate(pred)
    ce = -(truth * numpy.log(numpy.clip(pred, 1e-12, 1.0 - 1e-12))).sum(axis=1).mean()
    return float(ce), float((truth.argmax(1) == pred.argmax(1)).mean())


def main():
    session = open_session()
    words = vocabulary.get_classes(wanted_only=True, extend_reversed=False)
    proc, cfg, labels = make_processor(words)
    proc.summary()
    feed_train = utils.data_gen(proc, session, batch_size=BATCH, mode='training', pseudo_frequency=0.6)
    feed_val = utils.data_gen(proc, session, batch_size=BATCH, mode='validation', pseudo_frequency=0.0)
    net = models.speech_model('conv_1d_time_sliced_with_attention', cfg['desired_samples'],
                              num_classes=cfg['label_count'], **cfg)
    val_steps = proc.set_size('validation') // BATCH
    watchers = [
        validation_callbacks.ConfusionMatrixCallback(feed_val, val_steps, wanted_words=labels, all_words=labels,
                                                     label2int=proc.word_to_index),
        kc.ReduceLROnPlateau(monitor='val_categorical_accuracy', mode='max', factor=0.5, patience=4, verbose=1,
                             min_lr=1e-5),
        kc.TensorBoard(log_dir='tb_scalars'),
        kc.ModelCheckpoint('saved/ep-{epoch:03d}-vl-{val_loss:.4f}.hdf5', monitor='val_categorical_accuracy',
                           mode='max', save_best_only=True),
    ]
    history = net.fit_generator(feed_train, steps_per_epoch=proc.set_size('training') // BATCH, epochs=EPOCHS,
                                verbose=1, callbacks=watchers)
    scores = net.evaluate_generator(feed_val, val_steps)
    ce, acc = recompute_validation(net, proc, session, val_steps)
    import os
    rank = os.environ.get('RANK')
    with open('train_result.json' if rank is None else 'train_result_rank%s.json' % rank, 'w') as f:
        json.dump({'evaluate': [float(v) for v in scores], 'recomputed_val_loss': ce, 'recomputed_val_acc': acc,
                   'history_keys': sorted(history.history.keys()), 'n_labels': len(labels)}, f)


if __name__ == '__main__':
    sys.exit(main())
'''

PREDICT_CALLER = r'''
import glob
import json

import numpy
import keras.backend
import keras.models
from keras.activations import softmax
from keras.applications.mobilenet import DepthwiseConv2D

import classes as vocabulary
import input_data
import model as models
import utils


def load_clip(proc, session, cfg, path):
    feed = {proc.wav_filename_placeholder_: path,
            proc.foreground_volume_placeholder_: 1.0,
            proc.time_shift_placeholder_: 0,
            proc.background_volume_placeholder_: 0.0,
            proc.background_data_placeholder_: numpy.zeros((cfg['desired_samples'], 1))}
    return session.run(proc.background_clamp_, feed_dict=feed).flatten()


def main():
    session = keras.backend.get_session()
    keras.backend.set_learning_phase(0)
    words = vocabulary.get_classes(wanted_only=True)
    names = vocabulary.get_int2label(wanted_only=True)
    cfg = models.prepare_model_settings(len(input_data.prepare_words_list(words)), 16000, 1000, 25.0, 15.0, 80, 60,
                                        output_representation='raw')
    proc = input_data.AudioProcessor(['data/train/audio'], 12.0, 5.0, words, 10.0, 0.0, cfg, output_representation='raw')
    newest = sorted(glob.glob('saved/*.hdf5'))[-1]
    net = keras.models.load_model(newest, custom_objects={
        'relu6': models.relu6, 'overlapping_time_slice_stack': models.overlapping_time_slice_stack,
        'DepthwiseConv2D': DepthwiseConv2D, 'softmax': softmax, '<lambda>': utils.smooth_categorical_crossentropy})
    paths = sorted(glob.glob('data/train/audio/yes/*.wav'))[:40]
    batch = numpy.float32([load_clip(proc, session, cfg, p) for p in paths])
    plain = net.predict(batch)
    shifted = net.predict(numpy.roll(batch, -1500, axis=1))
    louder = net.predict(1.2 * batch)
    mean3 = (plain + louder + shifted) / 3
    numpy.save('tta_probs.npy', mean3)
    numpy.save('tta_terms.npy', numpy.stack([plain, louder, shifted]))
    numpy.save('tta_batch.npy', batch)
    with open('predict_result.json', 'w') as f:
        json.dump({'checkpoint': newest, 'labels': [names[int(i)] for i in mean3.argmax(axis=-1)]}, f)


if __name__ == '__main__':
    main()
'''


def _write_wav(path, x):
    from speech_recognition_amd.input_data import save_wav_file
    save_wav_file(str(path), x, 16000)


@pytest.fixture(scope="module")
def dataset(tmp_path_factory):
    root = tmp_path_factory.mktemp("kws_data")
    rng = np.random.RandomState(0)
    words = 'stop down off right up go on yes left no bed cat'.split()
    t = np.arange(16000) / 16000.0
    for ci, w in enumerate(words):
        d = root / 'data' / 'train' / 'audio' / w
        d.mkdir(parents=True)
        for i in range(28):
            x = 0.2 * np.sin(2 * np.pi * (180.0 * (ci + 1)) * t) + 0.05 * rng.randn(16000)
            n = rng.randint(12000, 16001)            # ragged lengths: DecodeWav pads to desired_samples
            _write_wav(d / ('%08x_nohash_%d.wav' % (rng.randint(0, 2 ** 31 - 1), i % 3)), x[:n])
    nd = root / 'data' / 'train' / 'audio' / '_background_noise_'
    nd.mkdir()
    _write_wav(nd / 'white.wav', 0.1 * rng.randn(16000 * 5))
    for ci, w in enumerate(words[:10]):
        d = root / 'data' / 'heng_pseudo' / w
        d.mkdir(parents=True)
        for i in range(6):
            x = 0.2 * np.sin(2 * np.pi * (180.0 * (ci + 1)) * t) + 0.05 * rng.randn(16000)
            _write_wav(d / ('clip_%03d.wav' % i), x)
    (root / 'train_caller.py').write_text(TRAIN_CALLER)
    (root / 'predict_caller.py').write_text(PREDICT_CALLER)
    return root


def _run(root, script, repo_root):
    env = dict(os.environ, PYTHONPATH=repo_root)
    r = subprocess.run([sys.executable, '-m', 'speech_recognition_amd.run_script', script], cwd=str(root), env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    out = r.stdout.decode(errors='replace')
    assert r.returncode == 0, out[-4000:]
    return out


def test_training_caller_runs_through_the_dropin_modules_and_learns(dataset, repo_root):
    out = _run(dataset, 'train_caller.py', repo_root)
    assert 'There are 13 classes.' in out            # AudioProcessor.summary(): 12 words + silence
    logs = [json.loads(l) for l in open(dataset / 'tb_scalars' / 'scalars.jsonl')]
    assert len(logs) == 3
    for k in ('loss', 'categorical_accuracy', 'val_loss', 'val_categorical_accuracy', 'lr',
              'val_mean_categorical_accuracy_all', 'val_mean_categorical_accuracy_wanted'):
        assert k in logs[-1], k
    assert logs[-1]['loss'] < logs[0]['loss']          # the tone task is learnable
    assert logs[-1]['categorical_accuracy'] > 0.5
    assert abs(logs[-1]['lr'] - 1e-3) < 1e-9           # 3 epochs: ReduceLROnPlateau (patience 4) has not fired
    res = json.load(open(dataset / 'train_result.json'))
    assert res['n_labels'] == 12
    assert res['history_keys'] == sorted(k for k in logs[-1] if k not in ('step', 'wall_time'))   # History saw val_* and lr
    # VALUES of the validation callback (row a16): an independent pass over the validation partition with the final
    # weights must reproduce what ConfusionMatrixCallback logged for the last epoch
    assert abs(res['recomputed_val_loss'] - logs[-1]['val_loss']) < 1e-5
    assert abs(res['recomputed_val_acc'] - logs[-1]['val_categorical_accuracy']) < 1e-9
    # evaluate_generator: Keras loss (smoothed CE + L2) and accuracy on the same batches
    assert abs(res['evaluate'][1] - logs[-1]['val_categorical_accuracy']) < 1e-9 and res['evaluate'][0] > 0
    text = open(dataset / 'confusion_matrix.txt').read()
    assert text.count('val_categorical_accuracy') == 3 and 'Predicted' in text
    assert ("[002]: val_categorical_accuracy: %.2f" % logs[-1]['val_categorical_accuracy']) in text
    ck = sorted(os.listdir(dataset / 'saved'))
    assert ck and ck[0].startswith('ep-00') and ck[0].endswith('.hdf5')
    best = max(l['val_categorical_accuracy'] for l in logs)
    first_best = [l for l in logs if l['val_categorical_accuracy'] == best][0]
    assert ('vl-%.4f' % first_best['val_loss']) in ck[-1]          # save_best_only on val_categorical_accuracy


def test_prediction_caller_tta_inference(dataset, repo_root):
    if not os.path.isdir(dataset / 'saved'):
        pytest.skip("training test did not run")
    _run(dataset, 'predict_caller.py', repo_root)
    probs = np.load(dataset / 'tta_probs.npy')
    terms = np.load(dataset / 'tta_terms.npy')
    batch = np.load(dataset / 'tta_batch.npy')
    assert probs.shape == (28, 12) and batch.shape == (28, 16000)       # the 28 'yes' wavs of the generated dataset
    np.testing.assert_allclose(probs.sum(axis=1), 1.0, rtol=1e-4)
    np.testing.assert_allclose(probs, (terms[0] + terms[1] + terms[2]) / 3, rtol=1e-6)
    # the clips the feed keys produced are the wav files themselves (DecodeWav scale, zero padded)
    from speech_recognition_amd.input_data import load_wav_file
    fns = sorted((dataset / 'data' / 'train' / 'audio' / 'yes').glob('*.wav'))
    for i in (0, 13, 27):
        w = load_wav_file(str(fns[i]))
        ref = np.zeros(16000, np.float32)
        ref[:len(w)] = w[:16000]
        assert np.array_equal(batch[i], ref)
    res = json.load(open(dataset / 'predict_result.json'))
    # (after ~60 training steps the BatchNorm moving averages - momentum 0.99 - are still far from the batch statistics,
    # so inference-mode predictions are not yet meaningful: the labels are checked for consistency, not for accuracy)
    from speech_recognition_amd.classes import get_int2label
    names = get_int2label(wanted_only=True)
    assert res['labels'] == [names[int(i)] for i in probs.argmax(axis=-1)] and len(res['labels']) == 28


def test_training_caller_on_two_data_parallel_ranks(dataset, repo_root, tmp_path):
    """The same caller under torch.distributed.run with two ranks (both on the one GPU over gloo - the hooks
    KWS_ONE_DEVICE; an N-GPU node runs RCCL): gradients are all-reduced every step, replicas are
    synchronised at fit start and at every epoch end, and ONLY rank 0 writes the checkpoints, the scalar log and the
    confusion-matrix files.  Both ranks must end with the same validation numbers."""
    import shutil
    work = tmp_path / "dp"
    work.mkdir()
    os.symlink(str(dataset / 'data'), str(work / 'data'))
    shutil.copy(str(dataset / 'train_caller.py'), str(work / 'train_caller.py'))
    env = dict(os.environ, PYTHONPATH=repo_root, KWS_ONE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', '29547', '-m', 'speech_recognition_amd.run_script', 'train_caller.py']
    r = subprocess.run(cmd, cwd=str(work), env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    assert r.returncode == 0, r.stdout.decode(errors='replace')[-4000:]
    logs = [json.loads(l) for l in open(work / 'tb_scalars' / 'scalars.jsonl')]
    assert len(logs) == 3                                  # one line per epoch: rank 1 did not write
    text = open(work / 'confusion_matrix.txt').read()
    assert text.count('val_categorical_accuracy') == 3
    r0 = json.load(open(work / 'train_result_rank0.json'))
    r1 = json.load(open(work / 'train_result_rank1.json'))
    assert r0['recomputed_val_acc'] == r1['recomputed_val_acc']
    assert abs(r0['recomputed_val_loss'] - r1['recomputed_val_loss']) < 1e-7          # replicas are identical
    assert abs(r0['recomputed_val_loss'] - logs[-1]['val_loss']) < 1e-5
    assert logs[-1]['loss'] < logs[0]['loss']
    assert len(os.listdir(work / 'saved')) >= 1
