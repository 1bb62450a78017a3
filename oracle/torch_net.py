"""Oracle, torch-CPU twin of oracle/net.py:TimeSlicedAttentionNet (reference model.py:775-838).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Same Keras-named parameters, same counter-based
dropout masks, same Keras optimizers and BatchNorm moving-average rule as the NumPy oracle, but the
layer arithmetic runs through torch.nn.functional on the HOST cores (oneDNN / MKL) with autograd for
the backward pass.  Two uses:

  * an independent implementation the hand-written NumPy forward/backward is cross-checked against
    (tests/test_oracle_net.py) - and, being ~20x faster than the NumPy loops, the CPU side of the
    val-acc parity run (scripts/val_acc_parity.py; reference loop train.py:56-75);
  * the "model step on CPU" leg of bench.py's cpu_baseline: the closest
    stand-in available here for the reference's TF-CPU path.

It never touches a GPU and the product never imports it.
"""
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

from . import layers as L

# NEGATIVE CONTROLS (round 6; scripts/val_acc_parity.py, tests/test_val_acc_gpu.py): deliberately WRONG backward passes of this twin,
# to show that the parity bars would catch a wrong gradient.  Never the default; the product knows nothing of them.
#   'bn_c2'   BatchNorm backward without its xhat * mean(g xhat) term (every BatchNorm layer)
#   'dw_flip' depthwise convolution: the input gradient uses the taps in reverse order (a correlation / convolution mix-up)
#   'pw_half' the weight gradient of ONE pointwise layer (conv1d_6) x 0.5 - invisible to RMSprop by construction (the update
#             g / sqrt(mean g^2) does not change when a tensor's gradient is scaled); caught by the one-step gradient bars instead
#   'l2_off'  the L2 term dropped from the gradient (1e-5 |w|^2: ~1e-6 of a typical gradient entry - caught by the total-loss and
#             the updated-weights bars of the step tests)
MUTATIONS = ('bn_c2', 'dw_flip', 'pw_half', 'l2_off')


class _BNNoC2(torch.autograd.Function):
    """training-mode BatchNorm whose backward drops the xhat * mean(g * xhat) term (mutation 'bn_c2')"""

    @staticmethod
    def forward(ctx, h, g, b):
        mean = h.mean(dim=(0, 2), keepdim=True)
        var = h.var(dim=(0, 2), unbiased=False, keepdim=True)
        rstd = torch.rsqrt(var + 1e-3)
        xhat = (h - mean) * rstd
        ctx.save_for_backward(xhat, rstd, g)
        return xhat * g[None, :, None] + b[None, :, None]

    @staticmethod
    def backward(ctx, dout):
        xhat, rstd, g = ctx.saved_tensors
        dg = (dout * xhat).sum(dim=(0, 2))
        db = dout.sum(dim=(0, 2))
        dx = g[None, :, None] * rstd * (dout - dout.mean(dim=(0, 2), keepdim=True))        # ... - xhat * mean(dout * xhat): dropped
        return dx, dg, db


class _DWFlip(torch.autograd.Function):
    """depthwise conv1d (already padded input) whose INPUT gradient is computed with the taps reversed (mutation 'dw_flip')"""

    @staticmethod
    def forward(ctx, h, w, stride):
        ctx.save_for_backward(h, w)
        ctx.stride = stride
        return F.conv1d(h, w, stride=stride, groups=w.shape[0])

    @staticmethod
    def backward(ctx, dout):
        h, w = ctx.saved_tensors
        dh = torch.nn.grad.conv1d_input(h.shape, w.flip(-1), dout, stride=ctx.stride, groups=w.shape[0])
        dw = torch.nn.grad.conv1d_weight(h, w.shape, dout, stride=ctx.stride, groups=w.shape[0])
        return dh, dw, None


def forward(net, params, x, y, seed, step, training=True, state=None, dropout='oracle', batch_stats=None, mutation=None):
    """The network of oracle/net.py:TimeSlicedAttentionNet.forward written with torch ops (channels-first inside).

    net: the NumPy oracle object (layer table only); params: {Keras name: torch tensor}; x [B, 16000], y one-hot.
    training=False normalises with `state` (moving statistics).  dropout: 'oracle' = the counter-based masks shared
    with the device, 'torch' = torch's own generator (timing runs), None = off.
    Returns (probabilities, data loss, L2 loss); batch_stats (a dict) collects (mean, biased var) per BN layer."""
    B = x.shape[0]
    dt = x.dtype
    h = F.pad(x, (10, 10)).unfold(1, 40, 20).permute(0, 2, 1)           # frames [B, 40, 800]  (model.py:805)
    h = F.conv1d(h, params['conv1d_1/kernel'].permute(2, 1, 0), stride=2)

    def bn_relu6(h, idx):
        g = params['batch_normalization_%d/gamma' % idx]
        b = params['batch_normalization_%d/beta' % idx]
        if training:
            if batch_stats is not None:
                with torch.no_grad():
                    batch_stats[idx] = (h.mean(dim=(0, 2)), h.var(dim=(0, 2), unbiased=False))
            h = _BNNoC2.apply(h, g, b) if mutation == 'bn_c2' else F.batch_norm(h, None, None, g, b, training=True, eps=1e-3)
        else:
            mm = state['batch_normalization_%d/moving_mean' % idx]
            mv = state['batch_normalization_%d/moving_variance' % idx]
            h = F.batch_norm(h, mm, mv, g, b, training=False, eps=1e-3)
        return torch.clamp(h, 0, 6)
    h = bn_relu6(h, 1)
    for i, blk in enumerate(net.blocks):
        w = params['depthwise_conv2d_%d/depthwise_kernel' % (i + 1)].reshape(3, blk['cin'])
        if mutation == 'dw_flip' and training:
            h = _DWFlip.apply(F.pad(h, blk['pad']), w.t().unsqueeze(1), blk['stride'])
        else:
            h = F.conv1d(F.pad(h, blk['pad']), w.t().unsqueeze(1), stride=blk['stride'], groups=blk['cin'])
        Wp = params['conv1d_%d/kernel' % (i + 2)].reshape(blk['cin'], blk['cout'])
        h = F.conv1d(h, Wp.t().unsqueeze(2))
        h = bn_relu6(h, i + 2)
    a = h.permute(0, 2, 1)                                               # [B, T, C]
    T, C = net.T, net.C
    flat = a.reshape(B, T * C)
    if training and dropout == 'oracle':
        m1 = torch.from_numpy(L.dropout_mask(L.dropout_key(seed, step, 1), B * T * C, 0.6).reshape(B, T * C)).to(dt)
        m2 = torch.from_numpy(L.dropout_mask(L.dropout_key(seed, step, 2), B * 2 * C, 0.6).reshape(B, 2 * C)).to(dt)
        fd = flat * m1 / 0.6
    elif training and dropout == 'torch':
        fd = F.dropout(flat, 0.4)
    else:
        fd = flat
    att = torch.softmax(fd @ params['dense_1/kernel'] + params['dense_1/bias'], dim=1)
    xa = a * att[:, :, None]
    feat = torch.cat([xa.max(dim=1).values, a.mean(dim=1)], dim=1)
    if training and dropout == 'oracle':
        feat = feat * m2 / 0.6
    elif training and dropout == 'torch':
        feat = F.dropout(feat, 0.4)
    p = torch.softmax(feat @ params['dense_2/kernel'], dim=1)
    ysm = y * 0.9 + 0.1 / y.shape[1]
    logits = torch.log(torch.clamp(p, 1e-7, 1 - 1e-7))
    loss = -(ysm * torch.log_softmax(logits, dim=1)).sum(dim=1).mean()
    reg = sum(1e-5 * (v ** 2).sum() for k, v in params.items() if k.endswith('kernel'))
    return p, loss, reg


class TorchTimeSlicedNet(object):
    """Trainable torch-CPU twin: Keras RMSprop / SGD-momentum (oracle/layers.py rules) and BN moving averages."""

    def __init__(self, num_classes=12, dtype=torch.float32, numpy_net=None, threads=None, mutation=None):
        from .net import TimeSlicedAttentionNet
        assert mutation is None or mutation in MUTATIONS, mutation
        self.mutation = mutation            # a NEGATIVE CONTROL (see MUTATIONS): a deliberately wrong backward pass
        if threads:
            torch.set_num_threads(int(threads))
        self.np_net = numpy_net if numpy_net is not None else TimeSlicedAttentionNet(num_classes=num_classes,
                                                                                       dtype=np.float32)
        self.dtype = dtype
        self.params = OrderedDict((k, torch.tensor(np.asarray(v), dtype=dtype, requires_grad=True))
                                  for k, v in self.np_net.params.items())
        self.state = OrderedDict((k, torch.tensor(np.asarray(v), dtype=dtype)) for k, v in self.np_net.state.items())
        self.slots = None
        self.opt_kind = None

    def init_optimizer(self, kind='rmsprop'):
        self.opt_kind = kind
        self.slots = OrderedDict((k, torch.zeros_like(v)) for k, v in self.params.items())

    def predict(self, x):
        with torch.no_grad():
            xt = torch.as_tensor(np.asarray(x), dtype=self.dtype)
            y = torch.zeros((xt.shape[0], self.np_net.num_classes), dtype=self.dtype)
            p, _, _ = forward(self.np_net, self.params, xt, y, 0, 0, training=False, state=self.state)
        return p.numpy()

    def train_step(self, x, y_onehot, lr, seed=0, step=0, dropout='oracle'):
        """One Keras train_on_batch (oracle/net.py:train_step): returns (total loss, accuracy)."""
        xt = torch.as_tensor(np.asarray(x), dtype=self.dtype)
        yt = torch.as_tensor(np.asarray(y_onehot), dtype=self.dtype)
        for v in self.params.values():
            v.grad = None
        stats = {}
        p, loss, reg = forward(self.np_net, self.params, xt, yt, seed, step, training=True, dropout=dropout,
                               batch_stats=stats, mutation=self.mutation)
        (loss if self.mutation == 'l2_off' else loss + reg).backward()
        if self.mutation == 'pw_half':
            self.params['conv1d_6/kernel'].grad.mul_(0.5)
        with torch.no_grad():
            for k, v in self.params.items():
                g, a = v.grad, self.slots[k]
                if self.opt_kind == 'rmsprop':      # Keras RMSprop: rho .9, eps 1e-8 outside the sqrt
                    a.mul_(0.9).addcmul_(g, g, value=0.1)
                    v.sub_(lr * g / (torch.sqrt(torch.clamp(a, min=0)) + 1e-8))
                else:                               # Keras SGD(momentum .9): v = m v - lr g; p += v
                    a.mul_(0.9).sub_(lr * g)
                    v.add_(a)
            for idx, (mean, var) in stats.items():  # moving = 0.99 moving + 0.01 batch, biased variance
                self.state['batch_normalization_%d/moving_mean' % idx].mul_(0.99).add_(0.01 * mean)
                self.state['batch_normalization_%d/moving_variance' % idx].mul_(0.99).add_(0.01 * var)
            acc = float((p.argmax(dim=1) == yt.argmax(dim=1)).float().mean())
        self.last_data_loss = float(loss.detach())      # the batch's data loss alone (what the device's metrics row holds)
        return float((loss + reg).detach()), acc
