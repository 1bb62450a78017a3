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

Data (round 5): the tone dataset of SURVEY 8d made HARD enough not to saturate - class c = 0.0774 N(0,1) + TONE_AMP sin(2 pi
(400 + 25 c) t) with TONE_AMP = 0.010 (the reference-statistics task used 0.05 and frequencies 200 Hz apart: both sides reached
1.000 on it, which a net with a modest gradient bug would also have done - VERDICT r4), and 10 % of ALL index entries (training,
pseudo, validation) relabelled with a random word, so the validation accuracy that can be reached is 0.937 (measured: both sides
settle there for tone amplitudes 0.007 ... 0.015) and how fast the TRAINING accuracy climbs under the same batches and dropout
masks (0.51, 0.79, 0.85, 0.87, 0.90 ... per epoch at 0.010; 0.59, 0.84, 0.88 ... at 0.015) depends on the gradients being right.  Built on the device by bench.build_synthetic with a fixed seed; TWO sampler seeds.
Training is chaotic across ReLU6 kinks, so the two runs are not expected to agree weight by weight after hundreds of steps; what
must agree is the SETTLED validation accuracy (median of the last three epochs; tolerance TOL_SETTLED on the mean over seeds) -
and both must have learned the task.  Validation runs in inference mode on the BatchNorm MOVING statistics (momentum 0.99,
SURVEY D.2), which lag the batch statistics for the first few hundred steps: both sides validate at chance until then.
Both sides take the same FIXED schedule (1e-3, then one halving per epoch over the last three epochs: ReduceLROnPlateau's
factor) - a 10-epoch run never waits out the reference's patience of 4 (train.py:62-63) - and each side's own validation-
accuracy series is then REPLAYED through the product's ReduceLROnPlateau (keras_api.py: Keras 2.1.2's rule, pinned by fixture K3;
monitor val_categorical_accuracy, mode max, factor 0.5, patience 2 so that a 10-epoch series can fire) to show where the
reference's schedule would have fired on either side (`lr_replay`).

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

TOL_VAL_ACC = 0.05      # per seed: |settled val_acc(device) - settled val_acc(cpu)|
TOL_SETTLED = 0.02      # mean over the seeds of the same difference
TONE_AMP, TONE_STEP_HZ, LABEL_NOISE = 0.010, 25.0, 0.10
SEEDS = (4321, 97)
# the bars tests/test_val_acc_gpu.py asserts on every seed (one place, so that the negative controls are judged by the SAME bars)
BAR_TRAIN_CURVE = (0.05, 0.03)   # |train_acc(device) - train_acc(cpu)| per epoch: the steep first three epochs, then the rest
BAR_VAL_LOSS = 0.06              # |median of the last three epochs' validation loss, device - cpu|


def tripped_bars(dev, cpu, n_epochs=None, settled=True):
    """Which of the parity bars a (device series, CPU series) pair breaks; series = {"train_acc", "val_acc", "val_loss"} per epoch.
    n_epochs: compare the first n epochs only (a negative control's short run against the head of the device run);
    settled=False leaves out the two bars on the settled epochs (validation runs on the BatchNorm MOVING statistics, which sit at
    chance for the first ~4 epochs on both sides: a 4-epoch run has no settled epochs to compare)."""
    n = n_epochs or min(len(dev["train_acc"]), len(cpu["train_acc"]))
    out = []
    for e in range(n):
        d = abs(dev["train_acc"][e] - cpu["train_acc"][e])
        if not d < (BAR_TRAIN_CURVE[0] if e < 3 else BAR_TRAIN_CURVE[1]):
            out.append({"bar": "train_curve", "epoch": e, "difference": d, "allowed": BAR_TRAIN_CURVE[0] if e < 3 else BAR_TRAIN_CURVE[1]})
    if settled:
        med = lambda a: float(np.median(a[n - 3:n]))
        d = abs(med(dev["val_acc"]) - med(cpu["val_acc"]))
        if not d <= TOL_SETTLED:
            out.append({"bar": "settled_val_acc", "difference": d, "allowed": TOL_SETTLED})
        d = abs(med(dev["val_loss"]) - med(cpu["val_loss"]))
        if not d < BAR_VAL_LOSS:
            out.append({"bar": "val_loss", "difference": d, "allowed": BAR_VAL_LOSS})
    return out


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


def replay_reduce_lr(series, patience=2, factor=0.5, base_lr=1e-3, min_lr=1e-5):
    """the product's ReduceLROnPlateau (Keras 2.1.2's rule) fed one side's validation-accuracy series: the epochs after
    which it halves the learning rate, and the lr series it would have set (train.py:62-63 with a patience a short run can reach)"""
    from speech_recognition_amd.keras_api import ReduceLROnPlateau

    class _Lr(object):
        def __init__(self, v): self.value = np.float32(v)

    class _Opt(object):
        def __init__(self, v): self.lr = _Lr(v)

    class _Model(object):
        def __init__(self, v): self.optimizer = _Opt(v)
    cb = ReduceLROnPlateau(monitor='val_categorical_accuracy', mode='max', factor=factor, patience=patience, min_lr=min_lr)
    cb.model = _Model(base_lr)
    cb.on_train_begin()
    fired, lrs = [], []
    for e, v in enumerate(series):
        before = float(cb.model.optimizer.lr.value)
        cb.on_epoch_end(e, {'val_categorical_accuracy': float(v)})
        after = float(cb.model.optimizer.lr.value)
        if after < before:
            fired.append(e)
        lrs.append(after)
    return {"fired_after_epochs": fired, "lr": lrs}


def run_one(device, proc, settings, words, seed, epochs, steps, batch, val_batches, quiet, cpu_threads, negative_controls=(),
            nc_epochs=4):
    import bench
    from speech_recognition_amd.keras_api import Callback
    from oracle.net import TimeSlicedAttentionNet
    from oracle.torch_net import TorchTimeSlicedNet
    from speech_recognition_amd.callbacks import ConfusionMatrixCallback
    from speech_recognition_amd.model import speech_model
    from speech_recognition_amd.utils import data_gen
    np.random.seed(seed)
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

        class FirstEpochSteps(Callback):          # the per-step data loss of epoch 0 (the device's metrics ring, read once at epoch end)
            def __init__(self):
                Callback.__init__(self)
                self.loss = []

            def on_epoch_end(self, epoch, logs=None):
                if epoch == 0:
                    self.loss = [float(v) / batch for v in self.model._ring[:steps, 0].cpu().numpy()]
        first = FirstEpochSteps()
        t0 = time.time()
        hist = model.fit_generator(train, steps_per_epoch=steps, epochs=epochs, verbose=0, callbacks=[Schedule(), first, cb],
                                   max_queue_size=1)
        torch.cuda.synchronize()
        t_dev = time.time() - t0
    finally:
        os.chdir(cwd)
        sys.stdout = old_stdout
    dev_acc = [float(v) for v in hist.history['val_categorical_accuracy']]
    dev_loss = [float(v) for v in hist.history['val_loss']]
    dev_train_acc = [float(v) for v in hist.history['categorical_accuracy']]
    dev_train_loss = [float(v) for v in hist.history['loss']]
    # the enqueuer thread may have pulled batches past the last step: the twin trains on the first epochs*steps only
    batches = train.batches[:epochs * steps]
    # the validation generator keeps advancing from epoch to epoch (callbacks.py:63-66 calls next() validation_steps
    # times per epoch): the twin scores, epoch by epoch, the very batches the device callback consumed
    def run_twin(n_epochs, mutation=None, stop_when_tripped=None):
        """the CPU twin over the first n_epochs of the recorded batches (the lr schedule of the FULL run); a NEGATIVE CONTROL
        (mutation != None: a deliberately wrong backward pass, oracle/torch_net.py MUTATIONS) stops at the first epoch whose
        series breaks one of the bars against `stop_when_tripped` (the device's series)"""
        twin = TorchTimeSlicedNet(numpy_net=ora, threads=cpu_threads or min(os.cpu_count() or 1, 16), mutation=mutation)
        twin.init_optimizer('rmsprop')
        ser = {"val_acc": [], "val_loss": [], "train_acc": [], "train_loss": [], "first_steps_data_loss": []}
        for e in range(n_epochs):
            accs, losses = [], []
            for s in range(steps):
                k = e * steps + s
                X, y = batches[k]
                l, a = twin.train_step(X, y, float(np.float32(lr_of_epoch(e, epochs))), seed=model.seed, step=k)
                accs.append(a)
                losses.append(l)
                if e == 0:
                    ser["first_steps_data_loss"].append(twin.last_data_loss)
            ser["train_acc"].append(float(np.mean(accs)))
            ser["train_loss"].append(float(np.mean(losses)))       # data loss (label smoothing 0.1) + L2, as Keras logs it
            vb = val.batches[e * val_batches:(e + 1) * val_batches]
            assert len(vb) == val_batches, "the device callback consumed fewer validation batches than expected"
            p = np.concatenate([twin.predict(X) for X, _ in vb])
            yt = np.concatenate([y for _, y in vb])
            ser["val_acc"].append(float((p.argmax(1) == yt.argmax(1)).mean()))
            ser["val_loss"].append(float(-(yt * np.log(np.clip(p, 1e-12, 1 - 1e-12))).sum(axis=1).mean()))
            if stop_when_tripped is not None and tripped_bars(stop_when_tripped, ser, n_epochs=e + 1, settled=False):
                break
        return ser
    t0 = time.time()
    cpu_ser = run_twin(epochs)
    cpu_acc, cpu_loss, cpu_train_acc = cpu_ser["val_acc"], cpu_ser["val_loss"], cpu_ser["train_acc"]
    t_cpu = time.time() - t0
    # NEGATIVE CONTROLS: the same twin with a deliberately wrong backward pass, on the same batches, judged against
    # the DEVICE's series by the same bars the parity test asserts - a suite whose bars no wrong gradient can break proves nothing
    dev_ser = {"val_acc": dev_acc, "val_loss": dev_loss, "train_acc": dev_train_acc, "train_loss": dev_train_loss,
               "first_steps_data_loss": first.loss}
    controls = {}
    for mut in negative_controls:
        t0 = time.time()
        ser = run_twin(min(nc_epochs, epochs), mutation=mut, stop_when_tripped=dev_ser)
        n = len(ser["train_acc"])
        controls[mut] = {"epochs_run": n, "tripped": tripped_bars(dev_ser, ser, n_epochs=n, settled=False), "train_acc": ser["train_acc"],
                         "train_loss": ser["train_loss"], "device_train_loss": dev_train_loss[:n], "first_steps_data_loss": ser["first_steps_data_loss"], "unmutated_twin_train_loss": cpu_ser["train_loss"][:n],
                         "val_loss": ser["val_loss"], "device_train_acc": dev_train_acc[:n],
                         "unmutated_twin_train_acc": cpu_train_acc[:n], "seconds": time.time() - t0}
    # "settled" = the median of the last three epochs: with Keras' BatchNorm momentum of 0.99 the inference-mode accuracy of
    # either side can sit lower for a single late epoch (and torch-CPU's threaded reductions are not run-to-run deterministic)
    settled = lambda a: float(np.median(a[-3:]))
    return {"seed": int(seed), "val_acc_settled": settled(dev_acc), "val_acc_cpu_settled": settled(cpu_acc),
            "val_acc_best": max(dev_acc), "val_acc_cpu_best": max(cpu_acc), "val_acc_last": dev_acc[-1], "val_acc_cpu_last": cpu_acc[-1],
            "device": {"val_acc": dev_acc, "val_loss": dev_loss, "train_acc": dev_train_acc, "train_loss": dev_train_loss,
                       "first_steps_data_loss": first.loss, "seconds": t_dev,
                       "lr_replay": replay_reduce_lr(dev_acc)},
            "cpu": {"val_acc": cpu_acc, "val_loss": cpu_loss, "train_acc": cpu_train_acc, "train_loss": cpu_ser["train_loss"],
                    "first_steps_data_loss": cpu_ser["first_steps_data_loss"], "seconds": t_cpu,
                    "lr_replay": replay_reduce_lr(cpu_acc),
                    "what": "oracle/torch_net.py (torch-CPU f32), same batches, same dropout masks, same RMSprop"},
            "tripped": tripped_bars(dev_ser, cpu_ser),
            "negative_controls": controls}


def run(device=None, epochs=10, steps=100, batch=64, val_batches=None, bank=16384, quiet=False, cpu_threads=None, seeds=SEEDS,
        tone_amp=TONE_AMP, tone_step_hz=TONE_STEP_HZ, label_noise=LABEL_NOISE, negative_controls=(), nc_epochs=4):
    """negative_controls: mutations of oracle/torch_net.py MUTATIONS run on the FIRST seed only (nc_epochs epochs each, stopping at
    the first epoch that breaks a bar)"""
    import bench
    device = device if device is not None else torch.device("cuda", 0)
    spec = bench.build_synthetic(device, bank, seed=59185, tone_amp=tone_amp, tone_step_hz=tone_step_hz, label_noise=label_noise)
    # every epoch scores the WHOLE validation partition (the generator walks it in index order, wanted words first: a part of it
    # per epoch makes the series a function of which part was scored - measured: 0.96 / 0.97 on the unknown-word quarters, 0.2 - 0.9 on
    # the others, period 4)
    if val_batches is None:
        val_batches = len(spec['index']['validation']) // batch
    from speech_recognition_amd.input_data import AudioProcessor, prepare_words_list
    from speech_recognition_amd.model import prepare_model_settings
    words = prepare_words_list(bench.WANTED)
    settings = prepare_model_settings(label_count=len(words), sample_rate=16000, clip_duration_ms=1000,
                                      window_size_ms=30.0, window_stride_ms=10.0, dct_coefficient_count=80,
                                      num_log_mel_features=60, output_representation='raw')
    # ONE processor (clip bank, generator stream) for all seeds: a seed is the sampler's RNG state and fresh generators
    proc = AudioProcessor(spec, 13.0, 60.0, bench.WANTED, 10.0, 0.0, settings, output_representation='raw', device=device)
    per_seed = [run_one(device, proc, settings, words, sd, epochs, steps, batch, val_batches, quiet, cpu_threads,
                        negative_controls=negative_controls if (i == 0 or os.environ.get("KWS_NC_ALL_SEEDS")) else (), nc_epochs=nc_epochs)
                for i, sd in enumerate(seeds)]
    mean = lambda k: float(np.mean([r[k] for r in per_seed]))
    d_settled = [r["val_acc_settled"] - r["val_acc_cpu_settled"] for r in per_seed]
    res = {"val_acc": mean("val_acc_last"), "val_acc_cpu": mean("val_acc_cpu_last"), "val_acc_best": mean("val_acc_best"),
           "val_acc_cpu_best": mean("val_acc_cpu_best"), "val_acc_settled": mean("val_acc_settled"),
           "val_acc_cpu_settled": mean("val_acc_cpu_settled"),
           "val_acc_parity": {"tolerance": TOL_VAL_ACC, "tolerance_mean_settled": TOL_SETTLED,
                              "task": {"tone_amp": tone_amp, "tone_step_hz": tone_step_hz, "label_noise": label_noise, "noise_std": 0.0774,
                                       "bank_clips": bank, "what": "class c = noise + tone_amp sin(2 pi (400 + step c) t); a share "
                                                                   "label_noise of all index entries relabelled with a random word"},
                              "seeds": [int(sd) for sd in seeds],
                              "settled_device_minus_cpu": d_settled, "mean_abs_settled_difference": float(abs(np.mean(d_settled))),
                              "ok_settled": bool(abs(np.mean(d_settled)) <= TOL_SETTLED and all(abs(d) <= TOL_VAL_ACC for d in d_settled)),
                              "ok": bool(abs(mean("val_acc_last") - mean("val_acc_cpu_last")) <= TOL_VAL_ACC),
                              "ok_best": bool(abs(mean("val_acc_best") - mean("val_acc_cpu_best")) <= TOL_VAL_ACC),
                              "lr_replay_same_epochs": [r["device"]["lr_replay"]["fired_after_epochs"] == r["cpu"]["lr_replay"]["fired_after_epochs"]
                                                        for r in per_seed],
                              "negative_controls": {
                                  "what": "the CPU twin with a deliberately WRONG backward pass (oracle/torch_net.py MUTATIONS) on the first "
                                          "seed's batches, judged against the device's series by the bars of the parity test "
                                          "(train-curve %g / %g per epoch; validation sits at chance for the first ~4 epochs on both sides, "
                                          "so a short control has no settled epochs)" % BAR_TRAIN_CURVE,
                                  "tripped": {m: [t["bar"] + "@%d" % t.get("epoch", -1) for t in c["tripped"]]
                                              for m, c in per_seed[0]["negative_controls"].items()}},
                              "epochs": epochs, "steps_per_epoch": steps, "batch": batch,
                              "validation_clips": val_batches * batch,
                              "validation_rows_disjoint_from_training": not (
                                  set(r for r, _ in spec['index']['validation']) &
                                  set(r for r, _ in spec['index']['training'] + spec['index']['pseudo'])),
                              "per_seed": per_seed}}
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--epochs", type=int, default=10)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--val-batches", type=int, default=None)
    ap.add_argument("--tone-amp", type=float, default=TONE_AMP)
    ap.add_argument("--seeds", default=",".join(str(v) for v in SEEDS))
    ap.add_argument("--json", default=None)
    ap.add_argument("--negative-controls", default="", help="comma list of oracle/torch_net.py MUTATIONS (or 'all')")
    ap.add_argument("--nc-epochs", type=int, default=4)
    a = ap.parse_args()
    from oracle.torch_net import MUTATIONS
    ncs = MUTATIONS if a.negative_controls == "all" else tuple(m for m in a.negative_controls.split(",") if m)
    if not torch.cuda.is_available():
        raise SystemExit("val_acc_parity needs an MI355X for the device side")
    res = run(epochs=a.epochs, steps=a.steps, batch=a.batch, val_batches=a.val_batches, tone_amp=a.tone_amp,
              seeds=tuple(int(v) for v in a.seeds.split(",")), negative_controls=ncs, nc_epochs=a.nc_epochs)
    txt = json.dumps(res, indent=1)
    print(txt)
    if a.json:
        with open(a.json, "w") as f:
            f.write(txt)
    return 0 if res["val_acc_parity"]["ok_settled"] else 1


if __name__ == "__main__":
    sys.exit(main())
