"""GPU parity of the whole network program (forward, loss, backward, optimizer, BN moving stats)
against the CPU oracle.  Bar from BASELINE.json north_star: class indices bit-exact, softmax
within 1e-3 (we check 2e-5).

Gradients: the network is piecewise smooth - a pre-activation within f32 rounding of a ReLU6 kink
(0 or 6), or two max-pool candidates within rounding of each other, may legitimately fall on
different sides in the f32 device pass and the f64 oracle, and the gradient jumps there (measured:
one flipped element moves upstream gradients by 3e-3..3e-2 of their max).  The tests therefore read
the device's own discrete decisions (ReLU6 masks, max-pool winners) back through
kws_net_debug_view and hand them to the oracle's backward pass; with decisions aligned every one of
the 51 gradient tensors has to match to 5e-5 of its max."""
import numpy as np
import pytest
import torch

from oracle import layers as OL
from oracle.net import TimeSlicedAttentionNet
from speech_recognition_amd import _lib
from speech_recognition_amd.net import DeviceNet

pytestmark = pytest.mark.gpu


def _oracle(num_classes=12, seed=5, **kw):
    ora = TimeSlicedAttentionNet(num_classes=num_classes, dtype=np.float64, **kw)
    rng = np.random.RandomState(seed)
    for k in ora.params:                      # de-trivialise BN affine / bias so their grads matter
        if k.endswith('gamma'):
            ora.params[k] = (1.0 + 0.1 * rng.randn(*ora.params[k].shape)).astype(np.float32)
        if k.endswith('beta') or k.endswith('bias'):
            ora.params[k] = (0.1 * rng.randn(*ora.params[k].shape)).astype(np.float32)
    for k in ora.state:
        if k.endswith('moving_mean'):
            ora.state[k] = (0.05 * rng.randn(*ora.state[k].shape)).astype(np.float32)
        else:
            ora.state[k] = (1.0 + 0.2 * rng.rand(*ora.state[k].shape)).astype(np.float32)
    return ora


def _pair(num_classes=12, seed=5, **kw):
    ora = _oracle(num_classes, seed, **kw)
    net = DeviceNet(_lib.KWS_NET_TS_ATTENTION, num_classes, **kw)
    net.set_weights(dict(ora.params, **ora.state))
    return ora, net


def _batch(B, num_classes, seed, L=16000):
    rng = np.random.RandomState(seed)
    t = np.arange(L) / 16000.0
    lab = rng.randint(0, num_classes, B)
    x = rng.randn(B, L) * 0.0774 + 0.05 * np.sin(2 * np.pi * 200.0 * (1 + lab)[:, None] * t[None])
    return x.astype(np.float32), np.eye(num_classes, dtype=np.float32)[lab]


def _device_decisions(net, ora, B):
    """ReLU6 masks of the 12 BN layers and the max-pool winners, recomputed from the device's own
    pre-BN tensors / BN tables / attention weights with the kernel's f32 arithmetic."""
    masks = {}
    pre12 = None
    shapes = [(B, ora.blocks[0]['Lin'], ora.blocks[0]['cin'])] + [(B, b['Lout'], b['cout']) for b in ora.blocks]
    for l in range(12):
        y = net.debug_view(B, 0, l).reshape(shapes[l])
        bn = net.debug_view(B, 2, l)
        C = shapes[l][2]
        # fmaf(y, scale, shift) rounded once to f32
        pre = (y.astype(np.float64) * bn[:C].astype(np.float64) + bn[C:2 * C].astype(np.float64)).astype(np.float32)
        masks[l + 1] = ((pre > 0) & (pre <= 6)).astype(np.float64)
        pre12 = pre
    x12 = np.minimum(np.maximum(pre12, np.float32(0)), np.float32(6))
    att = net.debug_view(B, 3, 0).reshape(B, -1)
    xa = x12 * att[:, :, None]                      # f32 product, as in ts_tail_kernel
    ind = (xa == xa.max(axis=1, keepdims=True)).astype(np.float64)
    return masks, ind


def _check_grads(ora, net, x, y, seed, step, B, tol=5e-5, **kw):
    masks, ind = _device_decisions(net, ora, B)
    loss, p, grads, cache = ora.loss_and_grads(x.astype(np.float64), y.astype(np.float64), seed=seed, step=step,
                                               relu_masks=masks, pool_ind=ind, **kw)
    g = net.grads_dict()
    flips = sum(int((masks[i] != OL.relu6_mask(cache['bn%d' % i][3])).sum()) for i in range(1, 13))
    # the decisions handed over are the oracle's own but for pre-activations within f32 rounding of a kink: a device whose
    # pre-activation PATTERN were wrong would show here, whatever the aligned gradients say
    assert flips <= max(2, 1e-4 * sum(m.size for m in masks.values())), "kink flips: %d" % flips
    for k, ref in grads.items():
        if k in ora.l2_names:                   # the HIP path folds L2 into the optimizer
            ref = ref - 2e-5 * ora.params[k].astype(np.float64)
        ref = ref.reshape(g[k].shape)
        err = np.abs(g[k] - ref).max() / max(np.abs(ref).max(), 1e-7)
        assert err < tol, (k, err, "kink flips: %d" % flips)
    return loss, p, grads, cache


def test_tensor_table_matches_keras_names_and_shapes():
    ora, net = _pair()
    assert net.count_params() == 1198601 and net.trainable_count() == 1191433
    for k, v in list(ora.params.items()) + list(ora.state.items()):
        assert net.tensors[k].shape == v.shape, k
    assert [s.name for s in net.tensors.values() if not s.is_state] == list(ora.params.keys())
    w = net.get_weights()
    for k, v in ora.params.items():
        assert np.array_equal(w[k], v)


def test_predict_matches_oracle():
    ora, net = _pair()
    x, _ = _batch(9, 12, 1)
    p = net.predict(torch.from_numpy(x).cuda()).cpu().numpy()
    ref = ora.forward(x.astype(np.float64), training=False)
    assert np.abs(p - ref).max() < 1e-5          # north_star bar: 1e-3
    assert np.array_equal(p.argmax(1), ref.argmax(1))


@pytest.mark.parametrize("B", [3, 6, 37])
def test_train_fwd_bwd_matches_oracle(B):
    ora, net = _pair()
    x, y = _batch(B, 12, B)
    probs = net.train_fwd_bwd(torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda(), seed=1234567, step=3)
    torch.cuda.synchronize()
    loss, p, grads, cache = _check_grads(ora, net, x, y, 1234567, 3, B)
    got = probs.cpu().numpy()
    assert np.abs(got - p).max() < 2e-5          # north_star bar on softmax: 1e-3
    assert np.array_equal(got.argmax(1), p.argmax(1))   # class indices bit-exact
    m = net.metrics.cpu().numpy()
    assert abs(m[0] / B - loss) < 2e-5
    assert m[1] == (p.argmax(1) == y.argmax(1)).sum()
    # BN moving statistics were updated with the batch moments
    w = net.get_weights()
    for idx, (mean, var) in cache['batch_stats'].items():
        mm = ora.state['batch_normalization_%d/moving_mean' % idx].astype(np.float64)
        mv = ora.state['batch_normalization_%d/moving_variance' % idx].astype(np.float64)
        np.testing.assert_allclose(w['batch_normalization_%d/moving_mean' % idx], mm - (mm - mean) * 0.01, atol=2e-6)
        np.testing.assert_allclose(w['batch_normalization_%d/moving_variance' % idx], mv - (mv - var) * 0.01, rtol=2e-5)


@pytest.mark.parametrize("opt,B,n_steps", [("sgd", 8, 4), ("rmsprop", 8, 4), ("sgd", 64, 5)])
def test_training_steps_teacher_forced(opt, B, n_steps):
    """Consecutive train_on_batch steps (BASELINE configs[0] / C1: Keras SGD momentum .9 at ITS batch of 64; reference
    model.py:834: RMSprop 1e-3).  Training is chaotic across ReLU6 kinks and RMSprop's sign-like first
    steps, so the oracle is re-synchronised to the device weights before every step; what must hold
    at EVERY step: softmax 2e-5 (bar 1e-3), identical class indices, total loss (data + L2) 5e-5, all
    gradients (decision-aligned) 5e-5, BN moving statistics, and new weights = the Keras update rule
    applied to the device's own gradient."""
    ora, net = _pair()
    lr = 0.01 if opt == "sgd" else 1e-3
    names = [k for k in ora.params]
    for step in range(n_steps):
        w0 = net.get_weights()
        slots0 = net.slots.cpu().numpy()
        for k in names:
            ora.params[k] = w0[k].copy()
        for k in ora.state:
            ora.state[k] = w0[k].copy()
        x, y = _batch(B, 12, 100 + step)
        probs = net.train_fwd_bwd(torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda(), seed=99, step=step)
        reg = net.l2_loss().item()
        g_dev = net.grads_dict()
        loss, p, grads, cache = _check_grads(ora, net, x, y, 99, step, B)
        if opt == "sgd":
            net.sgd_step(lr, 0.9)
        else:
            net.rmsprop_step(lr)
        got = probs.cpu().numpy()
        assert np.abs(got - p).max() < 2e-5, step
        assert np.array_equal(got.argmax(1), p.argmax(1))
        m = net.metrics.cpu().numpy()
        assert abs(m[0] / B + reg - (loss + ora.reg_loss())) < 5e-5, step
        w1 = net.get_weights()
        for k in names:
            s = net.tensors[k]
            geff = g_dev[k].astype(np.float64) + 2.0 * s.l2 * w0[k].astype(np.float64)
            sl = slots0[s.offset:s.offset + s.size].reshape(s.shape).astype(np.float64)
            if opt == "sgd":
                ref, _ = OL.sgd_momentum_step(w0[k].astype(np.float64), geff, sl, lr, 0.9)
            else:
                ref, _ = OL.rmsprop_step(w0[k].astype(np.float64), geff, sl, lr)
            np.testing.assert_allclose(w1[k], ref, rtol=0, atol=2e-6, err_msg=k)
        for idx, (mean, var) in cache['batch_stats'].items():
            mm = w0['batch_normalization_%d/moving_mean' % idx].astype(np.float64)
            np.testing.assert_allclose(w1['batch_normalization_%d/moving_mean' % idx], mm - (mm - mean) * 0.01,
                                       atol=2e-6)
    x, _ = _batch(16, 12, 777)
    for k in names:
        ora.params[k] = w1[k]
    for k in ora.state:
        ora.state[k] = w1[k]
    p = net.predict(torch.from_numpy(x).cuda()).cpu().numpy()
    ref = ora.forward(x.astype(np.float64), training=False)
    assert np.abs(p - ref).max() < 2e-5
    assert np.array_equal(p.argmax(1), ref.argmax(1))


def test_data_parallel_shard_semantics():
    """Rows [4,8) of a global batch of 8 run as their own shard (row_offset=4, loss_batch=8): dropout
    masks are those of the global rows, the loss gradient is scaled by 1/8, results are
    bit-reproducible run to run (fixed-order reductions everywhere)."""
    ora, net = _pair()
    x, y = _batch(8, 12, 5)
    dx, dy = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
    p1 = net.train_fwd_bwd(dx[4:], dy[4:], seed=3, step=0, row_offset=4, loss_batch=8).clone()
    g1 = net.grads.clone()
    p2 = net.train_fwd_bwd(dx[4:], dy[4:], seed=3, step=0, row_offset=4, loss_batch=8)
    assert torch.equal(p1, p2) and torch.equal(g1, net.grads)
    loss, p, grads, cache = _check_grads(ora, net, x[4:], y[4:], 3, 0, 4, drop_offset=4, loss_scale_B=8)
    assert np.abs(p1.cpu().numpy() - p).max() < 2e-5


def test_tta_inference_matches_oracle():
    """BASELINE config C5 / make_submission.py:120-146: (p + p_loud + p_left) / 3 then argmax, and the
    6-term speed-TTA sum divided by 10."""
    from speech_recognition_amd.keras_api import Model, RMSprop
    from speech_recognition_amd.tta import predict_tta
    ora, net = _pair()
    model = Model(net, RMSprop())
    x, _ = _batch(10, 12, 42)
    xs, _ = _batch(10, 12, 43)
    probs, amax = predict_tta(model, torch.from_numpy(x).cuda())
    f = lambda a: ora.forward(a.astype(np.float64), training=False)
    ref = (f(x) + f(OL.tta_transform(x, 2)) + f(OL.tta_transform(x, 1))) / 3
    assert np.abs(probs.cpu().numpy() - ref).max() < 1e-5
    assert np.array_equal(amax.cpu().numpy(), ref.argmax(1))
    probs6, amax6 = predict_tta(model, torch.from_numpy(x).cuda(), torch.from_numpy(xs).cuda())
    ref6 = (f(x) + f(OL.tta_transform(x, 2)) + f(OL.tta_transform(x, 1)) + f(xs) + f(OL.tta_transform(xs, 3)) +
            f(OL.tta_transform(xs, 4))) / 10
    assert np.abs(probs6.cpu().numpy() - ref6).max() < 1e-5
    assert np.array_equal(amax6.cpu().numpy(), ref6.argmax(1))


@pytest.mark.parametrize("B", [1, 50, 130])
def test_first_convolution_kernels_at_awkward_batch_sizes(B):
    """conv1.hip at row counts that are not multiples of its 64-row tiles / 32-row units and where its statistics rows
    (one per workgroup, up to 768) outnumber a tile-per-row count: the whole training step against the float64 oracle
    (B = 1: BatchNorm over a single clip's time steps; 50 x 399 and 130 x 399 rows end inside a tile and a unit)."""
    ora, net = _pair()
    x, y = _batch(B, 12, 100 + B)
    probs = net.train_fwd_bwd(torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda(), seed=5, step=1)
    torch.cuda.synchronize()
    loss, p, grads, cache = _check_grads(ora, net, x, y, 5, 1, B, tol=1e-4)
    got = probs.cpu().numpy()
    assert np.abs(got - p).max() < 2e-5
    assert np.array_equal(got.argmax(1), p.argmax(1))
    y0 = net.debug_view(B, 0, 0).reshape(B, 399, 128)            # the first convolution's own output, pre-BN
    ref0 = cache['bn1'][0]
    assert np.abs(y0 - ref0.reshape(y0.shape)).max() < 2e-5 * max(1.0, np.abs(ref0).max())


@pytest.mark.parametrize("split,mode", [(1, 0), (6, 0), (10, 0), (6, 1), (3, 1)])
def test_split_step_is_bit_identical_to_the_whole_step(split, mode):
    """kws_net_train_fwd_bwd_part (the two-call form that lets the data-parallel step all-reduce the late layers'
    gradients during the early layers' backward): parts 1 + 2 give the bits of the one-call step, and after part 1 the
    gradient buffer from kws_net_grad_ready_offset onward is already final."""
    ora, net = _pair()
    # mode 1 (the separate launches of rounds 1 - 3) at a batch with ragged last tiles: a part's gradients are final when it returns
    B = 6 if mode == 0 else 70
    if mode != 0:
        if net.gemm_mode == 2:
            pytest.skip("the fp16 x 2 re-run of this file keeps its own arithmetic arm")
        net.set_gemm_mode(mode)
    x, y = _batch(B, 12, 41)
    dx, dy = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
    p0 = net.train_fwd_bwd(dx, dy, seed=9, step=2, row_offset=3, loss_batch=2 * B).clone()
    g0, m0, st0 = net.grads.clone(), net.metrics.clone(), net.state.clone()
    net.set_weights(dict(ora.params, **ora.state))               # BN moving statistics back to the start
    assert net.num_blocks() == 11
    off = net.grad_ready_offset(split)
    assert 0 < off < net.n_params
    # the BatchNorm in front of block `split` (blocks count from 0: block i owns conv1d_{i+2} / batch_normalization_{i+2})
    assert off == net.tensors['batch_normalization_%d/gamma' % (split + 1)].offset
    p1 = net.train_fwd_bwd_part(1, split, dx, dy, seed=9, step=2, row_offset=3, loss_batch=2 * B)
    torch.cuda.synchronize()
    assert torch.equal(net.grads[off:], g0[off:])                # final before part 2 has run
    assert torch.equal(p1, p0) and torch.equal(net.metrics, m0)
    net.train_fwd_bwd_part(2, split, dx, dy, seed=9, step=2, row_offset=3, loss_batch=2 * B)
    torch.cuda.synchronize()
    assert torch.equal(net.grads, g0) and torch.equal(net.state, st0)
    with pytest.raises(_lib.KwsError):
        net.train_fwd_bwd_part(1, 11, dx, dy, seed=9, step=2)
    with pytest.raises(_lib.KwsError):
        net.grad_ready_offset(0)


@pytest.mark.parametrize("B", [6, 70, 200])
def test_paired_backward_launch_is_bit_identical_to_separate_launches(B):
    """Round 4: in gemm mode 0 a layer's input-gradient and weight-gradient GEMMs go out as ONE launch (gemm.hip
    gemm_dgrad_wgrad_kernel: the NN walk on the first blocks, one weight-gradient work item per block behind them); mode 1 is
    the schedule of rounds 1 - 3 (two launches).  Same code paths, same per-element arithmetic: every gradient, the
    probabilities, the metrics and the BatchNorm state must agree bit for bit.  (B = 70: several row tiles and a ragged last
    one in every layer, more than one M-split in the early ones.)"""
    ora, net = _pair()
    x, y = _batch(B, 12, 43)
    dx, dy = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
    mode0 = net.gemm_mode                      # (the fp16 x 2 re-run of this file starts every net in mode 2)
    net.set_gemm_mode(0)
    try:
        p0 = net.train_fwd_bwd(dx, dy, seed=5, step=1).clone()
        g0, m0, st0 = net.grads.clone(), net.metrics.clone(), net.state.clone()
        for mode in (1,):
            net.set_weights(dict(ora.params, **ora.state))
            net.set_gemm_mode(mode)
            p1 = net.train_fwd_bwd(dx, dy, seed=5, step=1)
            torch.cuda.synchronize()
            assert torch.equal(p1, p0) and torch.equal(net.metrics, m0) and torch.equal(net.state, st0), mode
            assert torch.equal(net.grads, g0), mode
    finally:
        net.set_gemm_mode(mode0)
    assert float(g0.abs().max()) > 0
    with pytest.raises(_lib.KwsError):         # round 5's third schedule (mode 3) left the library in round 6
        net.set_gemm_mode(3)
