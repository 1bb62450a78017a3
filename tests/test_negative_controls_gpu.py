"""NEGATIVE CONTROLS for the parity suite: the oracle is given a deliberately WRONG backward pass and the bars of
the existing step-level parity tests (tests/test_net_gpu.py: every gradient tensor within 5e-5 of its maximum against the oracle on
the device's own discrete decisions; total loss 5e-5; updated weights 2e-6) MUST break.  A suite that stayed green against a wrong
oracle would prove nothing about the device.  The mutations are those the review named, plus a tap-order mix-up:

  pw_half   the weight gradient of ONE pointwise layer x 0.5
  bn_c2     BatchNorm backward without its xhat * mean(g xhat) term
  l2_off    the L2 term dropped (gradient and loss)
  dw_flip   depthwise convolution: input gradient with the taps reversed

(The end-to-end val-acc run has its own controls in scripts/val_acc_parity.py; RMSprop's update g / sqrt(mean g^2) is invariant
to a per-tensor gradient scale and 1e-5 |w|^2 is ~1e-6 of a gradient entry, so `pw_half` and `l2_off` cannot bend a learning curve -
they are caught HERE.)"""
import numpy as np
import pytest
import torch

from oracle import layers as OL
from speech_recognition_amd import _lib

from test_net_gpu import _batch, _check_grads, _pair

pytestmark = pytest.mark.gpu


def _device_step(B=37, seed=1234567, step=3):
    ora, net = _pair()
    if net.gemm_mode == 2:
        pytest.skip("the fp16 x 2 re-run of the suite keeps its own tolerances")
    x, y = _batch(B, 12, B)
    net.train_fwd_bwd(torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda(), seed=seed, step=step)
    torch.cuda.synchronize()
    return ora, net, x, y, seed, step, B


def test_unmutated_oracle_passes_the_gradient_bar():
    ora, net, x, y, seed, step, B = _device_step()
    _check_grads(ora, net, x, y, seed, step, B)              # the positive control: same call, no mutation


def test_pointwise_weight_gradient_halved_trips_the_gradient_bar():
    ora, net, x, y, seed, step, B = _device_step()
    real = ora.loss_and_grads

    def mutated(*a, **kw):
        loss, p, grads, cache = real(*a, **kw)
        grads['conv1d_6/kernel'] = grads['conv1d_6/kernel'] * 0.5
        return loss, p, grads, cache
    ora.loss_and_grads = mutated
    with pytest.raises(AssertionError) as e:
        _check_grads(ora, net, x, y, seed, step, B)
    assert "conv1d_6/kernel" in str(e.value)


def test_batchnorm_backward_without_c2_trips_the_gradient_bar(monkeypatch):
    ora, net, x, y, seed, step, B = _device_step()

    def bn_bwd_no_c2(dout, yv, gamma, stats):
        mean, var, rstd = stats
        n = yv.shape[0] * yv.shape[1]
        xhat = (yv - mean) * rstd
        dbeta = dout.sum(axis=(0, 1))
        dgamma = (dout * xhat).sum(axis=(0, 1))
        return (gamma * rstd) * (dout - dbeta / n), dgamma, dbeta          # ... - xhat * (dgamma / n): dropped
    monkeypatch.setattr(OL, "bn_train_bwd", bn_bwd_no_c2)
    with pytest.raises(AssertionError):
        _check_grads(ora, net, x, y, seed, step, B)


def test_depthwise_input_gradient_with_reversed_taps_trips_the_gradient_bar(monkeypatch):
    ora, net, x, y, seed, step, B = _device_step()
    real = OL.dwconv_bwd

    def flipped(dy, xin, w, stride, pad):
        dx, _ = real(dy, xin, w[::-1], stride, pad)
        _, dw = real(dy, xin, w, stride, pad)
        return dx, dw
    monkeypatch.setattr(OL, "dwconv_bwd", flipped)
    with pytest.raises(AssertionError):
        _check_grads(ora, net, x, y, seed, step, B)


def test_l2_term_dropped_trips_the_loss_and_update_bars():
    """The product folds L2 into the optimizer kernel and reports it through kws_l2_loss: the bars that see it are the TOTAL loss
    (5e-5) and the updated weights (2e-6 against the Keras rule applied to gradient + 2 l2 w), tests/test_net_gpu.py
    test_training_steps_teacher_forced."""
    ora, net, x, y, seed, step, B = _device_step(B=8, seed=99, step=0)
    names = list(ora.params)
    w0 = net.get_weights()
    for k in names:
        ora.params[k] = w0[k].copy()
    # three warm steps so that RMSprop's accumulators are past their sign-like first updates
    dx, dy = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
    for s in range(3):
        net.train_fwd_bwd(dx, dy, seed=99, step=s)
        net.rmsprop_step(1e-3)
    w0 = net.get_weights()
    slots0 = net.slots.cpu().numpy()
    net.train_fwd_bwd(dx, dy, seed=99, step=3)
    reg = net.l2_loss().item()
    g_dev = net.grads_dict()
    m = net.metrics.cpu().numpy()
    net.rmsprop_step(1e-3)
    w1 = net.get_weights()
    for k in names:
        ora.params[k] = w0[k].copy()
    for k in ora.state:
        ora.state[k] = w0[k].copy()
    # (a) total loss: device data loss + device L2 against an oracle WITHOUT the L2 term
    loss = ora.loss_and_grads(x.astype(np.float64), y.astype(np.float64), seed=99, step=3)[0]
    assert abs(m[0] / B + reg - (loss + ora.reg_loss())) < 5e-4          # with the term: the bar's order of magnitude (kinks not aligned here)
    assert abs(m[0] / B + reg - loss) > 50 * 5e-5                         # without it: far outside the 5e-5 bar
    # (b) updated weights: the Keras rule on the device's gradient WITH the L2 term passes 2e-6, WITHOUT it breaks it
    worst_with, worst_without = 0.0, 0.0
    for k in names:
        s = net.tensors[k]
        assert (s.l2 == np.float32(1e-5)) == (k in ora.l2_names), k       # the device's table carries the reference's regularizers
        sl = slots0[s.offset:s.offset + s.size].reshape(s.shape).astype(np.float64)
        g = g_dev[k].astype(np.float64)
        ref_with, _ = OL.rmsprop_step(w0[k].astype(np.float64), g + 2.0 * s.l2 * w0[k].astype(np.float64), sl, 1e-3)
        ref_without, _ = OL.rmsprop_step(w0[k].astype(np.float64), g, sl, 1e-3)
        worst_with = max(worst_with, float(np.abs(w1[k] - ref_with).max()))
        worst_without = max(worst_without, float(np.abs(w1[k] - ref_without).max()))
    assert worst_with < 2e-6, worst_with
    assert worst_without > 2e-6, worst_without


# ---- the residual family (conv_1d_log_mfcc, BASELINE configs[2]): the same idea against tests/test_logmfcc_gpu.py's gradient bar (1e-4) ----
def _lm_device_step(B=19, nc=32):
    import test_logmfcc_gpu as T
    ora, net = T._pair(nc)
    x, y = T._batch(B, nc, B)
    net.train_fwd_bwd(torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda(), seed=77, step=2)
    torch.cuda.synchronize()
    masks, args = T._decisions(net, ora, B)
    return ora, net, x, y, masks, args


def _lm_worst_gradient_error(ora, net, x, y, masks, args):
    loss, p, grads, cache = ora.loss_and_grads(x.astype(np.float64), y.astype(np.float64), seed=77, step=2, relu_masks=masks, pool_args=args)
    g = net.grads_dict()
    worst = 0.0
    for k, ref in grads.items():
        if k in ora.l2_names:
            ref = ref - 2e-5 * ora.params[k].astype(np.float64)
        ref = ref.reshape(g[k].shape)
        worst = max(worst, float(np.abs(g[k] - ref).max() / max(np.abs(ref).max(), 1e-7)))
    return worst


def test_residual_net_unmutated_oracle_passes_and_mutations_trip_the_gradient_bar(monkeypatch):
    """Three wrong backward passes of the ORACLE's residual program must break the 1e-4 bar the unmutated one passes: the max-pool
    join routing the gradient to the LAST maximum of a window instead of the first (MaxPoolGrad's rule; the device kernels'
    `w1 > w0` test), the BatchNorm backward without its c2 term, and the last block's MAIN branch cut out of the join (only the
    shortcut path carries gradient upstream)."""
    import oracle.net as ON
    ora, net, x, y, masks, args = _lm_device_step()
    assert _lm_worst_gradient_error(ora, net, x, y, masks, args) < 1e-4          # the positive control
    # (1) pool routing: the winners handed over by the device, inverted
    flipped = {i: 1 - a for i, a in args.items()}
    assert _lm_worst_gradient_error(ora, net, x, y, masks, flipped) > 1e-2
    # (2) BatchNorm backward without the xhat * mean(g xhat) term
    real_bn = OL.bn_train_bwd

    def bn_bwd_no_c2(dout, yv, gamma, stats):
        mean, var, rstd = stats
        n = yv.shape[0] * yv.shape[1]
        xhat = (yv - mean) * rstd
        dbeta = dout.sum(axis=(0, 1))
        dgamma = (dout * xhat).sum(axis=(0, 1))
        return (gamma * rstd) * (dout - dbeta / n), dgamma, dbeta
    monkeypatch.setattr(OL, "bn_train_bwd", bn_bwd_no_c2)
    assert _lm_worst_gradient_error(ora, net, x, y, masks, args) > 1e-3
    monkeypatch.setattr(OL, "bn_train_bwd", real_bn)
    # (3) the residual join: the gradient that flows through the MAIN branch of the last block zeroed (the first max-pool backward of
    # the pass) - everything upstream of that block then sees the shortcut path alone
    real_pool = ON.maxpool_same_bwd
    calls = {"n": 0}

    def pool_bwd_first_call_zeroed(do, arg, pool, L):
        calls["n"] += 1
        out = real_pool(do, arg, pool, L)
        return out * 0.0 if calls["n"] == 1 else out
    monkeypatch.setattr(ON, "maxpool_same_bwd", pool_bwd_first_call_zeroed)
    assert _lm_worst_gradient_error(ora, net, x, y, masks, args) > 1e-2
