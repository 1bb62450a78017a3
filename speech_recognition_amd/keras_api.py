"""The slice of the Keras-2.1.2 training-loop surface that the reference's scripts touch
(train.py:56-75, make_submission.py:64-146, callbacks.py), re-hosted on the HIP network programs.

Host-side Python only - every numeric step is a libkws_hip.so call made by `DeviceNet`.
Semantics pinned by SURVEY.md Appendix D.6/D.7 (and fixtures K3/K4): `fit_generator` pulls batches
from ONE background thread through a depth-10 queue; epoch logs are batch-size-weighted means;
`ReduceLROnPlateau` / `ModelCheckpoint` follow Keras 2.1.2 line by line in behaviour;
`predict` defaults to 32-row mini-batches.
"""
from __future__ import division, print_function

import json
import os
import queue
import sys
import threading
import time
from collections import OrderedDict

import numpy as np
import torch

from . import _lib, parallel
from .device_array import DeviceArray, as_device_f32


def _inputs(x, device):
    """One f32 device tensor per batch.  A multi-input Keras model (conv_1d_mfcc_and_raw: `[mfcc, raw]`, the
    generator's 'mfcc_and_raw' output) reaches the network program as ONE row per clip, inputs side by side."""
    if isinstance(x, (list, tuple)):
        parts = [as_device_f32(v, device) for v in x]
        return parts[0] if len(parts) == 1 else torch.cat(parts, dim=1)
    return as_device_f32(x, device)


def _batch_len(x):
    return len(x[0]) if isinstance(x, (list, tuple)) else len(x)
from .net import DeviceNet


# ------------------------------------------------------------------------------------------------
# optimizers (hyper-parameters only; the update itself is kws_rmsprop_step / kws_sgd_momentum_step)
# ------------------------------------------------------------------------------------------------
class _LrVar(object):
    """Stand-in for the Keras backend variable `optimizer.lr` (read/written by ReduceLROnPlateau
    through K.get_value / K.set_value; Keras stores it as float32)."""

    def __init__(self, value):
        self.value = np.float32(value)

    def __float__(self):
        return float(self.value)


class Optimizer(object):
    def __init__(self, lr):
        self.lr = _LrVar(lr)


class RMSprop(Optimizer):
    """keras.optimizers.RMSprop(lr=0.001, rho=0.9, epsilon=1e-8, decay=0.) - SURVEY D.5."""

    def __init__(self, lr=0.001, rho=0.9, epsilon=1e-8, decay=0.0):
        Optimizer.__init__(self, lr)
        self.rho, self.epsilon, self.decay = rho, epsilon, decay
        if decay:
            raise NotImplementedError("RMSprop decay != 0 is not used by the reference")

    def apply(self, net, grad_scale, stream=None):
        net.rmsprop_step(float(self.lr), self.rho, self.epsilon, grad_scale, stream)


class SGD(Optimizer):
    """keras.optimizers.SGD(lr, momentum, nesterov=False) - reference model.py:96,110."""

    def __init__(self, lr=0.01, momentum=0.0, decay=0.0, nesterov=False):
        Optimizer.__init__(self, lr)
        self.momentum = momentum
        if decay or nesterov:
            raise NotImplementedError("SGD decay / nesterov are not used by the reference")

    def apply(self, net, grad_scale, stream=None):
        net.sgd_step(float(self.lr), self.momentum, grad_scale, stream)


# ------------------------------------------------------------------------------------------------
# callbacks
# ------------------------------------------------------------------------------------------------
class Callback(object):
    def __init__(self):
        self.validation_data = None
        self.model = None
        self.params = {}

    def set_params(self, params):
        self.params = params

    def set_model(self, model):
        self.model = model

    def on_train_begin(self, logs=None):
        pass

    def on_train_end(self, logs=None):
        pass

    def on_epoch_begin(self, epoch, logs=None):
        pass

    def on_epoch_end(self, epoch, logs=None):
        pass

    def on_batch_begin(self, batch, logs=None):
        pass

    def on_batch_end(self, batch, logs=None):
        pass


class History(Callback):
    def on_train_begin(self, logs=None):
        self.epoch = []
        self.history = {}

    def on_epoch_end(self, epoch, logs=None):
        self.epoch.append(epoch)
        for k, v in (logs or {}).items():
            self.history.setdefault(k, []).append(v)


class ReduceLROnPlateau(Callback):
    """Keras 2.1.2 ReduceLROnPlateau (replaying it on the reference's logged accuracies
    reproduces the logged lr series exactly - fixture K3)."""

    def __init__(self, monitor='val_loss', factor=0.1, patience=10, verbose=0, mode='auto', epsilon=1e-4,
                 cooldown=0, min_lr=0):
        Callback.__init__(self)
        if factor >= 1.0:
            raise ValueError('ReduceLROnPlateau does not support a factor >= 1.0.')
        self.monitor, self.factor, self.patience, self.verbose = monitor, factor, patience, verbose
        self.mode, self.epsilon, self.cooldown, self.min_lr = mode, epsilon, cooldown, min_lr
        self.cooldown_counter = 0
        self.wait = 0
        self.best = 0
        self.monitor_op = None
        self._reset()

    def _reset(self):
        if self.mode not in ('auto', 'min', 'max'):
            self.mode = 'auto'
        if self.mode == 'min' or (self.mode == 'auto' and 'acc' not in self.monitor):
            self.monitor_op = lambda a, b: np.less(a, b - self.epsilon)
            self.best = np.Inf if hasattr(np, 'Inf') else np.inf
        else:
            self.monitor_op = lambda a, b: np.greater(a, b + self.epsilon)
            self.best = -np.inf
        self.cooldown_counter = 0
        self.wait = 0
        self.lr_epsilon = self.min_lr * 1e-4

    def on_train_begin(self, logs=None):
        self._reset()

    def in_cooldown(self):
        return self.cooldown_counter > 0

    def on_epoch_end(self, epoch, logs=None):
        logs = logs if logs is not None else {}
        logs['lr'] = float(self.model.optimizer.lr.value)
        current = logs.get(self.monitor)
        if current is None:
            print('Reduce LR on plateau conditioned on metric `%s` which is not available. '
                  'Available metrics are: %s' % (self.monitor, ','.join(list(logs.keys()))), file=sys.stderr)
            return
        if self.in_cooldown():
            self.cooldown_counter -= 1
            self.wait = 0
        if self.monitor_op(current, self.best):
            self.best = current
            self.wait = 0
        elif not self.in_cooldown():
            if self.wait >= self.patience:
                old_lr = float(self.model.optimizer.lr.value)
                if old_lr > self.min_lr + self.lr_epsilon:
                    new_lr = max(old_lr * self.factor, self.min_lr)
                    self.model.optimizer.lr.value = np.float32(new_lr)
                    if self.verbose > 0:
                        print('\nEpoch %05d: reducing learning rate to %s.' % (epoch + 1, new_lr))
                    self.cooldown_counter = self.cooldown
                    self.wait = 0
            self.wait += 1


class ModelCheckpoint(Callback):
    """Keras 2.1.2 ModelCheckpoint: `filepath.format(epoch=epoch + 1, **logs)`.
    Weights are written as an .npz state-dict under Keras variable names (no h5py in the image)."""

    def __init__(self, filepath, monitor='val_loss', verbose=0, save_best_only=False, save_weights_only=False,
                 mode='auto', period=1):
        Callback.__init__(self)
        self.filepath, self.monitor, self.verbose = filepath, monitor, verbose
        self.save_best_only, self.save_weights_only, self.period = save_best_only, save_weights_only, period
        self.epochs_since_last_save = 0
        if mode == 'min' or (mode not in ('max',) and 'acc' not in monitor and not monitor.startswith('fmeasure')):
            self.monitor_op, self.best = np.less, np.inf
        else:
            self.monitor_op, self.best = np.greater, -np.inf

    def on_epoch_end(self, epoch, logs=None):
        logs = logs or {}
        self.epochs_since_last_save += 1
        if self.epochs_since_last_save < self.period:
            return
        self.epochs_since_last_save = 0
        filepath = self.filepath.format(epoch=epoch + 1, **logs)
        if self.save_best_only:
            current = logs.get(self.monitor)
            if current is None:
                print('Can save best model only with %s available, skipping.' % self.monitor, file=sys.stderr)
                return
            if not self.monitor_op(current, self.best):
                return
            if self.verbose > 0:
                print('\nEpoch %05d: %s improved from %0.5f to %0.5f, saving model to %s'
                      % (epoch + 1, self.monitor, self.best, current, filepath))
            self.best = current
        if parallel.rank() != 0:      # data-parallel: replicas are identical, rank 0 owns the files
            return
        d = os.path.dirname(filepath)
        if d and not os.path.isdir(d):
            os.makedirs(d)
        self.model.save(filepath)


class TensorBoard(Callback):
    """Scalar logger standing in for keras.callbacks.TensorBoard(log_dir) (train.py:64): one JSON
    line per epoch with the same scalar names the reference's event files hold."""

    def __init__(self, log_dir='./logs', **_ignored):
        Callback.__init__(self)
        self.log_dir = log_dir
        self._f = None

    def on_train_begin(self, logs=None):
        if parallel.rank() != 0:
            return
        if not os.path.isdir(self.log_dir):
            os.makedirs(self.log_dir)
        self._f = open(os.path.join(self.log_dir, 'scalars.jsonl'), 'a')

    def on_epoch_end(self, epoch, logs=None):
        if self._f is None:
            return
        rec = OrderedDict(step=epoch, wall_time=time.time())
        for k, v in sorted((logs or {}).items()):
            rec[k] = float(v)
        self._f.write(json.dumps(rec) + '\n')
        self._f.flush()

    def on_train_end(self, logs=None):
        if self._f:
            self._f.close()
            self._f = None


# ------------------------------------------------------------------------------------------------
# generator enqueuer: one daemon thread, queue depth 10 (Keras fit_generator defaults)
# ------------------------------------------------------------------------------------------------
class GeneratorEnqueuer(object):
    def __init__(self, generator, max_queue_size=10, device=None):
        self.generator = generator
        self.queue = queue.Queue(maxsize=max_queue_size)
        self._stop = threading.Event()
        self._thread = None
        self._device = device

    def start(self):
        def run():
            if self._device is not None:
                torch.cuda.set_device(self._device)
            try:
                while not self._stop.is_set():
                    item = next(self.generator)
                    while not self._stop.is_set():
                        try:
                            self.queue.put(item, timeout=0.05)
                            break
                        except queue.Full:
                            continue
            except StopIteration:
                self.queue.put(StopIteration)
            except BaseException as e:  # surface generator errors in the training thread
                self.queue.put(e)
        self._thread = threading.Thread(target=run, name="kws-generator-enqueuer")
        self._thread.daemon = True
        self._thread.start()

    def get(self):
        item = self.queue.get()
        if item is StopIteration:
            raise StopIteration
        if isinstance(item, BaseException):
            raise item
        return item

    def stop(self):
        self._stop.set()
        if self._thread is not None:
            # the producer may sit in queue.put() on a full queue (it polls the stop flag every 50 ms): free a slot so that it
            # returns at once - a fit_generator call otherwise pays up to 50 ms here (measured: 0.5 ms per step of a 100-step fit)
            deadline = time.time() + 10.0
            while self._thread.is_alive() and time.time() < deadline:
                try:
                    self.queue.get_nowait()
                except queue.Empty:
                    pass
                self._thread.join(timeout=0.002)


# ------------------------------------------------------------------------------------------------
# Model
# ------------------------------------------------------------------------------------------------
class Model(object):
    """Keras-shaped model around one DeviceNet replica (data-parallel when torch.distributed is
    initialised: local BN statistics, RCCL all-reduce(sum) of the flat gradient, /world)."""

    metrics_names = ['loss', 'categorical_accuracy']

    def __init__(self, net, optimizer, name='model', seed=87654321, loss='smooth_cce'):
        self.net = net
        self.loss = loss      # 'smooth_cce' (utils.py:87-108, model.py:835) or 'cce' (model.py:1477)
        self.optimizer = optimizer
        self.name = name
        self.stop_training = False
        self.seed = int(seed)
        self._step = 0
        self.history = None
        self._ring = torch.zeros((1024, 4), dtype=torch.float32, device=net.device)
        self._reg_every = 16
        self._reg_value = None
        self.device = net.device
        self.stream = torch.cuda.current_stream(net.device)
        # data-parallel gradient exchange: 0 = ONE all-reduce of the flat buffer after the backward pass; k >= 1 = the
        # gradients of blocks >= k (+ tail) are summed while the earlier blocks' backward still runs (bit-identical)
        self.allreduce_split = parallel.split_block_from_env()

    # -- weights -----------------------------------------------------------------------------------
    def count_params(self):
        return self.net.count_params()

    def get_weights(self):
        return list(self.net.get_weights().values())

    def set_weights(self, weights):
        names = list(self.net.tensors.keys())
        self.net.set_weights(OrderedDict(zip(names, weights)))

    def save_weights(self, filepath):
        w = self.net.get_weights()
        with open(filepath, 'wb') as f:
            np.savez(f, **{k.replace('/', '|'): v for k, v in w.items()})

    def save(self, filepath):
        """Weights + optimizer slots + lr under Keras variable names (.npz container)."""
        w = self.net.get_weights()
        blob = {k.replace('/', '|'): v for k, v in w.items()}
        blob['__optimizer_slots__'] = self.net.slots.cpu().numpy()
        blob['__lr__'] = np.float32(self.optimizer.lr.value)
        blob['__step__'] = np.int64(self._step)
        blob['__model_name__'] = np.array(self.name)
        blob['__num_classes__'] = np.int64(self.net.num_classes)
        blob['__input_size__'] = np.int64(self.net.input_size)
        blob['__spectrogram_length__'] = np.int64(self.net.spectrogram_length)
        blob['__num_features__'] = np.int64(self.net.num_features)
        with open(filepath, 'wb') as f:
            np.savez(f, **blob)

    def load_weights(self, filepath):
        with np.load(filepath) as z:
            w = OrderedDict((k.replace('|', '/'), z[k]) for k in z.files if not k.startswith('__'))
            self.net.set_weights(w)
            if '__optimizer_slots__' in z.files:
                self.net.slots.copy_(torch.from_numpy(z['__optimizer_slots__']))
                self.optimizer.lr.value = np.float32(z['__lr__'])
                self._step = int(z['__step__'])
        self.sync_replicas()

    def sync_replicas(self):
        """Data-parallel runs: every replica takes rank 0's parameters, BN moving statistics, optimizer slots, lr and
        step counter (no-op on one GPU).  Called at fit start, after load_weights and at every epoch end, so replicas
        cannot drift apart through a one-rank load or through their rank-local BN statistics."""
        if not parallel.active():
            return
        for buf in (self.net.params, self.net.state, self.net.slots):
            parallel.broadcast_params(buf)
        t = torch.tensor([float(self.optimizer.lr.value), float(self._step), float(self.stop_training)],
                         dtype=torch.float64, device=self.device if parallel.dist.get_backend() == 'nccl' else 'cpu')
        parallel.dist.broadcast(t, src=0)
        self.optimizer.lr.value = np.float32(t[0].item())
        self._step = int(t[1].item())
        self.stop_training = bool(t[2].item())

    def summary(self):
        print("Model: %s" % self.name)
        for s in self.net.tensors.values():
            print("  %-48s %-20s %d" % (s.name, s.shape, s.size))
        print("Total params: %d  (trainable %d)" % (self.net.count_params(), self.net.trainable_count()))

    # -- one step ------------------------------------------------------------------------------------
    def _train_step_async(self, x, y, metrics_row):
        """Enqueue forward+backward(+all-reduce)+optimizer for one batch; nothing is synchronised."""
        net = self.net
        xd = _inputs(x, self.device)
        yd = as_device_f32(y, self.device)
        world, rank = parallel.world_size(), parallel.rank()
        B = xd.shape[0]
        net.metrics = metrics_row
        split = self.allreduce_split if world > 1 and net.kind == _lib.KWS_NET_TS_ATTENTION else 0
        if split > 0:
            # the late layers' gradients (a contiguous tail of the flat buffer) are summed over the ranks while the early
            # layers' backward still runs; the two slices together are the one-buffer all-reduce, element for element
            off = net.grad_ready_offset(split)
            net.train_fwd_bwd_part(1, split, xd, yd, seed=self.seed, step=self._step, row_offset=rank * B, loss_batch=B * world)
            h = parallel.allreduce_begin(net.grads[off:])
            net.train_fwd_bwd_part(2, split, xd, yd, seed=self.seed, step=self._step, row_offset=rank * B, loss_batch=B * world)
            parallel.allreduce_grads(net.grads[:off])
            parallel.allreduce_wait(h)
        else:
            net.train_fwd_bwd(xd, yd, seed=self.seed, step=self._step, row_offset=rank * B, loss_batch=B * world)
            parallel.allreduce_grads(net.grads)   # RCCL sum over xGMI; grads are already scaled by 1/(B*world)
        self.optimizer.apply(net, 1.0)
        self._step += 1

    def train_on_batch(self, x, y):
        row = self._ring[0]
        self._train_step_async(x, y, row)
        reg = float(self.net.l2_loss().item())
        m = row.cpu().numpy()
        B = _batch_len(x)
        return [float(m[0]) / B + reg, float(m[1]) / B]

    def test_on_batch(self, x, y):
        p = self.predict_on_batch(x)
        yt = np.asarray(y, dtype=np.float64)
        if self.loss == 'cce':
            pn = p.astype(np.float64) / p.astype(np.float64).sum(axis=1, keepdims=True)
            data_loss = float((-(yt * np.log(np.clip(pn, 1e-7, 1 - 1e-7))).sum(axis=1)).mean())
        else:
            pc = np.clip(p.astype(np.float64), 1e-7, 1 - 1e-7)
            C = yt.shape[1]
            ysm = yt * 0.9 + 0.1 / C
            S = pc.sum(axis=1, keepdims=True)
            data_loss = float((-(ysm * (np.log(pc) - np.log(S))).sum(axis=1)).mean())
        loss = data_loss + float(self.net.l2_loss().item())
        acc = float((p.argmax(1) == yt.argmax(1)).mean())
        return [loss, acc]

    def predict_on_batch(self, x):
        xd = _inputs(x, self.device)
        return self.net.predict(xd).cpu().numpy()

    def predict(self, x, batch_size=32, verbose=0):
        """Keras default batch_size=32; results are independent of the chunking because
        inference uses moving statistics.  Device inputs are processed in large chunks."""
        xd = _inputs(x, self.device)
        n = xd.shape[0]
        chunk = max(int(batch_size), 1024)
        outs = [self.net.predict(xd[i:i + chunk].contiguous()) for i in range(0, n, chunk)]
        return torch.cat(outs, 0).cpu().numpy()

    def predict_device(self, x):
        return self.net.predict(_inputs(x, self.device))

    # -- loops ---------------------------------------------------------------------------------------
    def fit_generator(self, generator, steps_per_epoch, epochs=1, verbose=1, callbacks=None,
                      validation_data=None, validation_steps=None, class_weight=None, max_queue_size=10,
                      workers=1, use_multiprocessing=False, shuffle=True, initial_epoch=0):
        if workers != 1 or use_multiprocessing:
            raise NotImplementedError("the reference uses the Keras defaults workers=1, threads (SURVEY D.7)")
        self.history = History()
        cbs = list(callbacks or []) + [self.history]     # Keras 2.1.2 order: History last, so it records the val_* / lr
                                                         # entries the user callbacks inject into `logs`
        for cb in cbs:
            cb.set_model(self)
            cb.set_params({'epochs': epochs, 'steps': steps_per_epoch, 'verbose': verbose,
                           'metrics': self.metrics_names})
        if self._ring.shape[0] < steps_per_epoch:
            self._ring = torch.zeros((steps_per_epoch, 4), dtype=torch.float32, device=self.device)
        enq = GeneratorEnqueuer(generator, max_queue_size=max_queue_size, device=self.device)
        enq.start()
        self.stop_training = False
        try:
            self.sync_replicas()
            for cb in cbs:
                cb.on_train_begin()
            for epoch in range(initial_epoch, epochs):
                for cb in cbs:
                    cb.on_epoch_begin(epoch)
                t0 = time.time()
                sizes = []
                reg_sum, reg_n = 0.0, 0
                for step in range(steps_per_epoch):
                    x, y = enq.get()
                    for cb in cbs:
                        cb.on_batch_begin(step, {'batch': step, 'size': _batch_len(x)})
                    self._train_step_async(x, y, self._ring[step])
                    sizes.append(_batch_len(x))
                    if step % self._reg_every == 0:
                        reg_sum += float(self.net.l2_loss().item())   # also a natural sync point
                        reg_n += 1
                    if verbose and (step + 1) % max(1, steps_per_epoch // 10) == 0:
                        print("\rEpoch %d/%d  step %d/%d" % (epoch + 1, epochs, step + 1, steps_per_epoch), end='')
                        sys.stdout.flush()
                    for cb in cbs:
                        cb.on_batch_end(step, {'batch': step, 'size': _batch_len(x)})
                m = self._ring[:steps_per_epoch].cpu().numpy().astype(np.float64)
                # data-parallel: the logged training metrics are the GLOBAL batch's (sums over every rank's clips), the
                # same on all ranks - a rank-local value would differ per replica and mislead anything that reads `logs`
                loss_sum, acc_sum, n = parallel.allreduce_sums([m[:, 0].sum(), m[:, 1].sum(), float(sum(sizes))], self.device)
                logs = {'loss': loss_sum / n + reg_sum / max(reg_n, 1),
                        'categorical_accuracy': acc_sum / n}
                if parallel.active():
                    # rank 0's moving statistics become everyone's BEFORE validation: every rank then computes the
                    # same val_* logs, so ReduceLROnPlateau / early stopping decide alike on all replicas
                    parallel.broadcast_params(self.net.state)
                for cb in cbs:
                    cb.on_epoch_end(epoch, logs)
                if parallel.active():
                    self.sync_replicas()          # lr / stop flag as rank 0 decided them; also the barrier after file writes
                if verbose:
                    print("\rEpoch %d/%d - %.1fs - %s" % (epoch + 1, epochs, time.time() - t0,
                          " - ".join("%s: %.4f" % kv for kv in sorted(logs.items()))))
                if self.stop_training:
                    break
            for cb in cbs:
                cb.on_train_end()
        finally:
            enq.stop()
        return self.history

    def evaluate_generator(self, generator, steps, max_queue_size=10, workers=1, use_multiprocessing=False):
        tot, n = np.zeros(2), 0
        for _ in range(steps):
            x, y = next(generator)
            l, a = self.test_on_batch(x, y)
            tot += np.array([l, a]) * _batch_len(x)
            n += _batch_len(x)
        return list(tot / max(n, 1))

    def predict_generator(self, generator, steps, **_):
        return np.concatenate([self.predict_on_batch(next(generator)[0]) for _ in range(steps)], 0)
