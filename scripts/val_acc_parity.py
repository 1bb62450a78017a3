#!/usr/bin/env python
"""Val-acc parity (BASELINE.json metric, second half; SURVEY 8d "val-acc parity on the same synthetic data").

The reference's loop (train.py:56-75): fit_generator over data_gen('training') with ConfusionMatrixCallback running
the validation partition at every epoch end (callbacks.py:45-83).  Here that loop runs twice on the SAME batches:

  device  speech_model('conv_1d_time_sliced_with_attention') .fit_generator(...) through the product path
          (sampler -> kws_augment -> kws_net_train_fwd_bwd -> RMSprop), validation through ConfusionMatrixCallback;
  cpu     the oracle's torch-CPU twin (oracle/torch_net.py, pinned to the NumPy oracle in tests/test_oracle_net.py):
          same initial weights, the very batches the device generator produced (recorded on their way into
          fit_generator), the same counter-based dropout masks (seed, step), the same optimizer, the same validation
          clips.

Data: the tone dataset of SURVEY 8d (class c = 0.0774 N(0,1) + 0.05 sin(2 pi 200 (1 + c) t), clipped), built on the
device by bench.build_synthetic with a fixed seed.  Training is chaotic across ReLU6 kinks, so the two runs are not
expected to agree weight by weight after hundreds of steps; what must agree is the validation accuracy per epoch
(tolerance TOL_VAL_ACC) - and both must have learned the task.  Validation runs in inference mode on the BatchNorm
MOVING statistics (momentum 0.99): for the first ~300 steps they lag the batch statistics so far that
both runs validate at chance (measured: val_loss 2.487 = ln 12 on both sides after 120 steps while the training
accuracy is already 0.94; over 500 steps val_loss climbs in step on both sides - 2.49 / 2.51 / 2.59 / 2.83 device,
2.49 / 2.52 / 2.60 / 2.87 CPU - and then collapses to 0.55 within one epoch, which the two chaotic trajectories reach
an epoch apart), hence the default of 12 x 100 steps, by which both have converged.  The reference ends its runs on a
reduced learning rate (train.py:62-63: ReduceLROnPlateau(factor 0.5, patience 4); its logged series halve 1e-3 down to
6e-5, fixture K3); a 12-epoch run never waits out a patience of 4, so BOTH sides take the same fixed schedule here: 1e-3,
then one halving per epoch over the last three epochs.  With the weights slowing down the BatchNorm moving statistics
catch up and the inference-mode accuracy of either side stops hopping by a class between epochs (at a constant 1e-3 the
CPU twin ended repeated runs at 1.000 / 0.93 / 0.92 on the held-out partition).

The oracle is used here as the CHECKER (this script is measurement / test infrastructure, like bench.cpu_baseline).
usage:  python scripts/val_acc_parity.py [--epochs 3] [--steps 40] [--batch 64] [--json out.json]
"""
from __future__ import division, print_function

import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

TOL_VAL_ACC = 0.05      # |val_acc(device) - val_acc(cpu)| at the last epoch; measured differences are ~0.00-0.02


class Recorder(object):
    """Passes a generator's batches through and keeps host copies (what the CPU twin trains on)."""

    def __init__(self, gen, raw_of=lambda X: X):
        self.gen, self.raw_of, self.batches = gen, raw_of, []

    def __iter__(self):
        return self

    def __next__(self):
        X, y = next(self.gen)
        self.batches.append((np.asarray(self.raw_of(X), dtype=np.float32), np.asarray(y, dtype=np.float32)))
        return X, y

    next = __next__


def lr_of_epoch(epoch, epochs, base=1e-3, tail=3):
    """the schedule of both sides: base, then one halving per epoch over the last `tail` epochs (ReduceLROnPlateau's factor)"""
    return base * 0.5 ** max(0, epoch - (epochs - tail) + 1)


def run(device=None, epochs=12, steps=100, batch=64, val_batches=8, bank=4096, quiet=False, cpu_threads=None):
    import bench
    from speech_recognition_amd.keras_api import Callback
    from oracle.net import TimeSlicedAttentionNet
    from oracle.torch_net import TorchTimeSlicedNet
    from speech_recognition_amd.callbacks import ConfusionMatrixCallback
    from speech_recognition_amd.input_data import AudioProcessor, prepare_words_list
    from speech_recognition_amd.model import prepare_model_settings, speech_model
    from speech_recognition_amd.utils import data_gen
    device = device if device is not None else torch.device("cuda", 0)
    words = prepare_words_list(bench.WANTED)
    settings = prepare_model_settings(label_count=len(words), sample_rate=16000, clip_duration_ms=1000,
                                      window_size_ms=30.0, window_stride_ms=10.0, dct_coefficient_count=80,
                                      num_log_mel_features=60, output_representation='raw')
    spec = bench.build_synthetic(device, bank, seed=59185)
    proc = AudioProcessor(spec, 13.0, 60.0, bench.WANTED, 10.0, 0.0, settings, output_representation='raw', device=device)
    np.random.seed(4321)
    out_stream = sys.stderr if quiet else sys.stdout
    old_stdout, cwd = sys.stdout, os.getcwd()
    sys.stdout = out_stream
    tmp = tempfile.mkdtemp(prefix="kws_valacc_")
    os.chdir(tmp)                       # ConfusionMatrixCallback writes its two text files into the cwd
    try:
        train = Recorder(data_gen(proc, None, batch_size=batch, mode='training', pseudo_frequency=0.6))
        val = Recorder(data_gen(proc, None, batch_size=batch, mode='validation', pseudo_frequency=0.0))
        model = speech_model('conv_1d_time_sliced_with_attention', settings['desired_samples'],
                             num_classes=settings['label_count'])
        ora = TimeSlicedAttentionNet(num_classes=settings['label_count'], dtype=np.float32)
        model.net.set_weights(dict(ora.params, **ora.state))          # both runs start from the same weights
        cb = ConfusionMatrixCallback(val, val_batches, wanted_words=words, all_words=words, label2int=proc.word_to_index)

        class Schedule(Callback):
            def on_epoch_begin(self, epoch, logs=None):
                self.model.optimizer.lr.value = np.float32(lr_of_epoch(epoch, epochs))
        t0 = time.time()
        hist = model.fit_generator(train, steps_per_epoch=steps, epochs=epochs, verbose=0, callbacks=[Schedule(), cb],
                                   max_queue_size=1)
        torch.cuda.synchronize()
        t_dev = time.time() - t0
    finally:
        os.chdir(cwd)
        sys.stdout = old_stdout
    dev_acc = [float(v) for v in hist.history['val_categorical_accuracy']]
    dev_loss = [float(v) for v in hist.history['val_loss']]
    dev_train_acc = [float(v) for v in hist.history['categorical_accuracy']]
    # the enqueuer thread may have pulled batches past the last step: the twin trains on the first epochs*steps only
    batches = train.batches[:epochs * steps]
    # the validation generator keeps advancing from epoch to epoch (callbacks.py:63-66 calls next() validation_steps
    # times per epoch): the twin scores, epoch by epoch, the very batches the device callback consumed
    twin = TorchTimeSlicedNet(numpy_net=ora, threads=cpu_threads or min(os.cpu_count() or 1, 16))
    twin.init_optimizer('rmsprop')
    cpu_acc, cpu_loss, cpu_train_acc = [], [], []
    t0 = time.time()
    for e in range(epochs):
        accs = []
        for s in range(steps):
            k = e * steps + s
            X, y = batches[k]
            _, a = twin.train_step(X, y, float(np.float32(lr_of_epoch(e, epochs))), seed=model.seed, step=k)
            accs.append(a)
        cpu_train_acc.append(float(np.mean(accs)))
        vb = val.batches[e * val_batches:(e + 1) * val_batches]
        assert len(vb) == val_batches, "the device callback consumed fewer validation batches than expected"
        p = np.concatenate([twin.predict(X) for X, _ in vb])
        yt = np.concatenate([y for _, y in vb])
        cpu_acc.append(float((p.argmax(1) == yt.argmax(1)).mean()))
        cpu_loss.append(float(-(yt * np.log(np.clip(p, 1e-12, 1 - 1e-12))).sum(axis=1).mean()))
    t_cpu = time.time() - t0
    # final-epoch values (the parity bar) and the best epoch (what a save-best checkpoint keeps): with Keras' BatchNorm
    # momentum of 0.99 the inference-mode accuracy of BOTH sides still moves by a class (0.9 <-> 1.0) between late epochs
    # "settled" = the median of the last three epochs: a single late epoch of either side can sit a class lower (measured on
    # the CPU twin over repeated runs of the same batches: 1.000, 0.980, 0.879 in the last epoch - torch-CPU's threaded
    # reductions are not run-to-run deterministic and the trajectory is chaotic), the median of three does not
    settled = lambda a: float(np.median(a[-3:]))
    res = {"val_acc": dev_acc[-1], "val_acc_cpu": cpu_acc[-1], "val_acc_best": max(dev_acc), "val_acc_cpu_best": max(cpu_acc),
           "val_acc_settled": settled(dev_acc), "val_acc_cpu_settled": settled(cpu_acc),
           "val_acc_parity": {"tolerance": TOL_VAL_ACC, "ok": abs(dev_acc[-1] - cpu_acc[-1]) <= TOL_VAL_ACC,
                              "ok_best": abs(max(dev_acc) - max(cpu_acc)) <= TOL_VAL_ACC,
                              "ok_settled": abs(settled(dev_acc) - settled(cpu_acc)) <= TOL_VAL_ACC,
                              "epochs": epochs, "steps_per_epoch": steps, "batch": batch,
                              "validation_clips": val_batches * batch,
                              "validation_rows_disjoint_from_training": not (
                                  set(r for r, _ in spec['index']['validation']) &
                                  set(r for r, _ in spec['index']['training'] + spec['index']['pseudo'])),
                              "device": {"val_acc": dev_acc, "val_loss": dev_loss, "train_acc": dev_train_acc,
                                         "seconds": t_dev},
                              "cpu": {"val_acc": cpu_acc, "val_loss": cpu_loss, "train_acc": cpu_train_acc,
                                      "seconds": t_cpu, "what": "oracle/torch_net.py (torch-CPU f32), same batches, same "
                                                                "dropout masks, same RMSprop"}}}
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--epochs", type=int, default=12)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--val-batches", type=int, default=8)
    ap.add_argument("--json", default=None)
    a = ap.parse_args()
    if not torch.cuda.is_available():
        raise SystemExit("val_acc_parity needs an MI355X for the device side")
    res = run(epochs=a.epochs, steps=a.steps, batch=a.batch, val_batches=a.val_batches)
    txt = json.dumps(res, indent=1)
    print(txt)
    if a.json:
        with open(a.json, "w") as f:
            f.write(txt)
    return 0 if res["val_acc_parity"]["ok_settled"] else 1


if __name__ == "__main__":
    sys.exit(main())
