"""Cross-check of the oracle's hand-written forward/backward (oracle/net.py, oracle/layers.py)
against an independent implementation: torch CPU float64 ops + autograd.  This is a sanity
check of the restatement, not a reference pin (the reference ships no tests)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import layers as L
from oracle.net import TimeSlicedAttentionNet


def torch_forward(net, params, x, y, seed, step):
    """The same network through torch.nn.functional (oracle/torch_net.py, float64 here)."""
    from oracle.torch_net import forward
    return forward(net, params, torch.from_numpy(x).to(torch.float64), torch.from_numpy(y).to(torch.float64), seed, step)


def test_torch_twin_train_steps_follow_numpy_oracle():
    """oracle/torch_net.py:TorchTimeSlicedNet (the CPU trainer of the val-acc parity run and of bench.py's cpu_baseline)
    against the NumPy oracle's train_step: losses, parameters, BN moving statistics after 3 RMSprop / SGD steps, and the
    inference-mode probabilities that validation reads."""
    from oracle.torch_net import TorchTimeSlicedNet
    for kind, lr in (('rmsprop', 1e-3), ('sgd', 1e-2)):
        ora = TimeSlicedAttentionNet(dtype=np.float64)
        ora.init_optimizer(kind)
        twin = TorchTimeSlicedNet(dtype=torch.float64, numpy_net=TimeSlicedAttentionNet(dtype=np.float64))
        twin.init_optimizer(kind)
        rng = np.random.RandomState(5)
        for step in range(3):
            x = rng.randn(4, 16000) * 0.0774
            y = np.eye(12)[rng.randint(0, 12, 4)]
            l0, a0 = ora.train_step(x, y, lr, seed=3, step=step)
            l1, a1 = twin.train_step(x, y, lr, seed=3, step=step)
            assert abs(l0 - l1) < 1e-9 * max(1.0, abs(l0)) and a0 == a1, (kind, step, l0, l1)
        for k, v in ora.master.items():
            # RMSprop divides by sqrt(a): elements whose gradient is ~0 amplify rounding, hence the absolute floor
            np.testing.assert_allclose(twin.params[k].detach().numpy().reshape(v.shape), v, rtol=1e-6, atol=1e-7, err_msg=k)
        for k, v in ora.state.items():
            np.testing.assert_allclose(twin.state[k].numpy(), v, rtol=1e-9, atol=1e-12, err_msg=k)
        xv = rng.randn(3, 16000) * 0.0774
        np.testing.assert_allclose(twin.predict(xv), ora.forward(xv, training=False), rtol=1e-5, atol=1e-8)


def test_param_count_matches_reference_readme_and_survey():
    net = TimeSlicedAttentionNet()
    assert net.count_params() == 1198601          # SURVEY B.1; README.md:14 "roughly 1.250.000"
    assert sum(v.size for v in net.params.values()) == 1191433
    assert [b['Lout'] for b in net.blocks] == [397, 199, 197, 99, 97, 49, 47, 24, 22, 11, 9]
    assert net.blocks[9]['pad'] == (0, 1)          # right-biased SAME padding at L=22


def test_oracle_grads_match_torch_autograd():
    net = TimeSlicedAttentionNet(dtype=np.float64)
    rng = np.random.RandomState(3)
    # de-trivialise BN affine params so their gradients are exercised
    for k in net.params:
        if k.endswith('gamma'):
            net.params[k] = (1.0 + 0.1 * rng.randn(*net.params[k].shape)).astype(np.float32)
        if k.endswith('beta') or k.endswith('bias'):
            net.params[k] = (0.1 * rng.randn(*net.params[k].shape)).astype(np.float32)
    B = 3
    x = (rng.randn(B, 16000) * 0.0774).astype(np.float64)
    y = np.eye(12)[[2, 0, 7]]
    loss, p, grads, _ = net.loss_and_grads(x, y, seed=11, step=5)
    tparams = {k: torch.tensor(v.astype(np.float64), requires_grad=True) for k, v in net.params.items()}
    pt, tloss, treg = torch_forward(net, tparams, x, y, 11, 5)
    (tloss + treg).backward()
    np.testing.assert_allclose(p, pt.detach().numpy(), rtol=1e-9, atol=1e-12)
    assert abs(loss - tloss.item()) < 1e-10
    assert abs(net.reg_loss() - treg.item()) < 1e-10
    for k, g in grads.items():
        tg = tparams[k].grad.numpy().reshape(g.shape)
        scale = max(np.abs(tg).max(), 1e-12)
        assert np.abs(g - tg).max() / scale < 1e-8, k


def test_inference_uses_moving_stats():
    net = TimeSlicedAttentionNet(dtype=np.float64)
    x = np.random.RandomState(1).randn(2, 16000) * 0.05
    p1 = net.forward(x, training=False)
    p2 = net.forward(x[:1], training=False)
    np.testing.assert_allclose(p1[:1], p2, rtol=1e-10)
    np.testing.assert_allclose(p1.sum(axis=1), 1.0, rtol=1e-12)


# ---- a19: conv_1d_log_mfcc_model -------------------------------------------------------------------------
def _torch_logmfcc(net, params, x, y, seed, step):
    from oracle.net import LM_BLOCKS
    B = x.shape[0]
    dt = torch.float64

    def bn(h, idx, relu=True):
        g = params['batch_normalization_%d/gamma' % idx]
        b = params['batch_normalization_%d/beta' % idx]
        h = F.batch_norm(h, None, None, g, b, training=True, eps=1e-3)
        return torch.clamp(h, 0, 6) if relu else h

    def dwpw(h, dwn, pwn, cin, cout):
        w = params[dwn].reshape(3, cin)
        h = F.conv1d(F.pad(h, (1, 1)), w.t().unsqueeze(1), groups=cin)
        return F.conv1d(h, params[pwn].reshape(cin, cout).t().unsqueeze(2))
    h = torch.from_numpy(x).to(dt).reshape(B, net.T0, net.F).permute(0, 2, 1)
    h = bn(F.conv1d(h, params[net.first[0]].permute(2, 1, 0)), net.first[1])
    for blk in net.blocks:
        if 'short' in blk:
            res = bn(F.conv1d(h, params[blk['short'][0]].reshape(blk['cin'], blk['nf']).t().unsqueeze(2), stride=blk['stride']),
                     blk['short'][1], relu=False)
        else:
            res = h
        a = bn(dwpw(h, blk['dw1'], blk['pw1'], blk['cin'], blk['nf']), blk['bn1'])
        a = bn(dwpw(a, blk['dw2'], blk['pw2'], blk['nf'], blk['nf']), blk['bn2'])
        if blk['stride'] != 1:
            a = F.max_pool1d(a, blk['stride'], blk['stride'], ceil_mode=True)      # Keras padding='same'
        h = a + res
    u = bn(dwpw(h, net.att[0], net.att[1], net.C, 1), net.att[2])        # [B, 1, T]
    att = torch.softmax(u, dim=2)
    feat = (h * att).mean(dim=2)
    m = torch.from_numpy(L.dropout_mask(L.dropout_key(seed, step, 1), B * net.C, 0.8).reshape(B, net.C)).to(dt)
    p = torch.softmax((feat * m / 0.8) @ params['dense_1/kernel'] + params['dense_1/bias'], dim=1)
    yt = torch.from_numpy(y).to(dt)
    pn = p / p.sum(dim=1, keepdim=True)
    loss = -(yt * torch.log(torch.clamp(pn, 1e-7, 1 - 1e-7))).sum(dim=1).mean()
    reg = sum(1e-5 * (params[k] ** 2).sum() for k in net.l2_names)
    return p, loss, reg


def test_logmfcc_param_count_and_shapes():
    from oracle.net import LogMfccNet
    net = LogMfccNet(num_classes=32)
    assert net.count_params() == 784484
    assert (net.T, net.C) == (12, 256)
    assert [b['Lout'] for b in net.blocks] == [96, 96, 48, 48, 24, 24, 24, 12, 12, 12]


@pytest.mark.parametrize("T", [98, 65])
def test_logmfcc_grads_match_torch_autograd(T):
    from oracle.net import LogMfccNet
    net = LogMfccNet(num_classes=32, spectrogram_length=T, dtype=np.float64)
    rng = np.random.RandomState(4)
    for k in net.params:
        if k.endswith('gamma'):
            net.params[k] = (1.0 + 0.1 * rng.randn(*net.params[k].shape)).astype(np.float32)
        if k.endswith('beta') or k.endswith('bias'):
            net.params[k] = (0.1 * rng.randn(*net.params[k].shape)).astype(np.float32)
    B = 4
    x = (rng.randn(B, T * 40) * 3.0).astype(np.float64)
    y = np.eye(32)[[3, 0, 17, 31]]
    loss, p, grads, _ = net.loss_and_grads(x, y, seed=9, step=2)
    tparams = {k: torch.tensor(v.astype(np.float64), requires_grad=True) for k, v in net.params.items()}
    pt, tloss, treg = _torch_logmfcc(net, tparams, x, y, 9, 2)
    (tloss + treg).backward()
    np.testing.assert_allclose(p, pt.detach().numpy(), rtol=1e-9, atol=1e-12)
    assert abs(loss - tloss.item()) < 1e-10
    for k, g in grads.items():
        tg = tparams[k].grad.numpy().reshape(g.shape)
        assert np.abs(g - tg).max() / max(np.abs(tg).max(), 1e-12) < 1e-8, k


# ---- steffeNet --------------------------------------------------------------------------
def _torch_steffe(net, params, x, y, seed, step):
    B = x.shape[0]
    dt = torch.float64

    def bn(h, idx, relu=True):
        g = params['batch_normalization_%d/gamma' % idx]
        b = params['batch_normalization_%d/beta' % idx]
        h = F.batch_norm(h, None, None, g, b, training=True, eps=1e-3)
        return torch.clamp(h, 0, 6) if relu else h

    def dwpw(h, dwn, pwn, cin, cout, stride=1, pad=(1, 1)):
        w = params[dwn].reshape(3, cin)
        h = F.conv1d(F.pad(h, pad), w.t().unsqueeze(1), groups=cin, stride=stride)
        return F.conv1d(h, params[pwn].reshape(cin, cout).t().unsqueeze(2))
    h = torch.from_numpy(x).to(dt).reshape(B, 1, net.L_in)
    h = F.conv1d(F.pad(h, (net.pl0, net.pr0)), params[net.first[0]].permute(2, 1, 0), stride=net.S0)
    h = bn(h, net.first[1])
    h = bn(dwpw(h, net.ctx[0], net.ctx[1], net.C0, net.C0), net.ctx[2])
    for blk in net.blocks:
        if 'short' in blk:
            res = bn(F.conv1d(h, params[blk['short'][0]].reshape(blk['cin'], blk['nf']).t().unsqueeze(2), stride=blk['stride']),
                     blk['short'][1], relu=False)
        else:
            res = h
        a = bn(dwpw(h, blk['dw1'], blk['pw1'], blk['cin'], blk['nf'], blk['stride'], blk['pad1']), blk['bn1'])
        a = bn(dwpw(a, blk['dw2'], blk['pw2'], blk['nf'], blk['nf']), blk['bn2'])
        h = a + res
    feat = torch.cat([h.max(dim=2).values, h.mean(dim=2)], dim=1)
    m = torch.from_numpy(L.dropout_mask(L.dropout_key(seed, step, 1), B * 2 * net.C, 0.5).reshape(B, 2 * net.C)).to(dt)
    p = torch.softmax((feat * m / 0.5) @ params['dense_1/kernel'], dim=1)
    yt = torch.from_numpy(y).to(dt)
    ys = yt * 0.9 + 0.1 / yt.shape[1]                       # utils.py:87-108 smooth_categorical_crossentropy
    logits = torch.log(torch.clamp(p, 1e-7, 1 - 1e-7))
    loss = -(ys * torch.log_softmax(logits, dim=1)).sum(dim=1).mean()
    reg = sum(1e-5 * (params[k] ** 2).sum() for k in net.l2_names)
    return p, loss, reg


def test_steffenet_shapes_and_grads_match_torch_autograd():
    from oracle.net import SteffeNet
    full = SteffeNet(num_classes=12)
    assert [b['Lout'] for b in full.blocks] == [160, 160, 80, 80, 40, 40, 20, 20, 10, 10, 5, 5]
    assert (full.L0, full.pl0, full.pr0, full.T, full.C) == (320, 12, 13, 5, 1536)
    assert list(full.params)[:8] == ['conv1d_1/kernel', 'batch_normalization_1/gamma', 'batch_normalization_1/beta',
                                     'depthwise_conv2d_1/depthwise_kernel', 'conv1d_2/kernel',
                                     'batch_normalization_2/gamma', 'batch_normalization_2/beta', 'conv1d_3/kernel']
    assert 'dense_1/bias' not in full.params and 'conv1d_1/kernel' not in full.l2_names
    # small widths keep the float64 torch cross-check fast; same structure
    net = SteffeNet(num_classes=12, input_size=3200, filter_widths=[24, 32], c0=16, dtype=np.float64)
    rng = np.random.RandomState(6)
    for k in net.params:
        if k.endswith('gamma'):
            net.params[k] = (1.0 + 0.1 * rng.randn(*net.params[k].shape)).astype(np.float32)
        if k.endswith('beta'):
            net.params[k] = (0.1 * rng.randn(*net.params[k].shape)).astype(np.float32)
    B = 3
    x = (rng.randn(B, 3200) * 0.3).astype(np.float64)
    y = np.eye(12)[[3, 0, 11]]
    loss, p, grads, _ = net.loss_and_grads(x, y, seed=9, step=2)
    tparams = {k: torch.tensor(v.astype(np.float64), requires_grad=True) for k, v in net.params.items()}
    pt, tloss, treg = _torch_steffe(net, tparams, x, y, 9, 2)
    (tloss + treg).backward()
    np.testing.assert_allclose(p, pt.detach().numpy(), rtol=1e-9, atol=1e-12)
    assert abs(loss - tloss.item()) < 1e-10
    for k, g in grads.items():
        tg = tparams[k].grad.numpy().reshape(g.shape)
        assert np.abs(g - tg).max() / max(np.abs(tg).max(), 1e-12) < 1e-8, k


# ---- conv_1d_residual -------------------------------------------------------------------
def _torch_residual(net, params, x, y, seed, step):
    B = x.shape[0]
    dt = torch.float64

    def bn(h, idx, relu=True):
        g = params['batch_normalization_%d/gamma' % idx]
        b = params['batch_normalization_%d/beta' % idx]
        h = F.batch_norm(h, None, None, g, b, training=True, eps=1e-3)
        return torch.clamp(h, 0, 6) if relu else h

    def dwpw(h, dwn, pwn, cin, cout, stride=1, pad=(1, 1)):
        w = params[dwn].reshape(3, cin)
        h = F.conv1d(F.pad(h, pad), w.t().unsqueeze(1), groups=cin, stride=stride)
        return F.conv1d(h, params[pwn].reshape(cin, cout).t().unsqueeze(2))
    xt = torch.from_numpy(x).to(dt)
    frames = F.pad(xt, (10, 10)).unfold(1, 40, 20)                               # [B, 800, 40]
    h = bn(F.conv1d(frames.permute(0, 2, 1), params[net.first[0]].permute(2, 1, 0), stride=2), net.first[1])
    for blk in net.blocks:
        if 'short' in blk:
            res = bn(F.conv1d(h, params[blk['short'][0]].reshape(blk['cin'], blk['nf']).t().unsqueeze(2), stride=blk['stride']),
                     blk['short'][1], relu=False)
        else:
            res = h
        a = bn(dwpw(h, blk['dw1'], blk['pw1'], blk['cin'], blk['nf']), blk['bn1'])
        a = bn(dwpw(a, blk['dw2'], blk['pw2'], blk['nf'], blk['nf']), blk['bn2'])
        _, pl, pr = L.same_pad(blk['Lin'], 3, blk['stride'])
        a = F.max_pool1d(F.pad(a, (pl, pr), value=float('-inf')), 3, blk['stride'])
        h = a + res
    for r in net.red:
        h = bn(dwpw(h, r['dw'], r['pw'], r['cin'], r['cout'], r['stride'], r['pad']), r['bn'])
    feat = h.mean(dim=2)
    m = torch.from_numpy(L.dropout_mask(L.dropout_key(seed, step, 1), B * net.C, 0.5).reshape(B, net.C)).to(dt)
    p = torch.softmax((feat * m / 0.5) @ params['dense_1/kernel'] + params['dense_1/bias'], dim=1)
    yt = torch.from_numpy(y).to(dt)
    pn = p / p.sum(dim=1, keepdim=True)
    loss = -(yt * torch.log(torch.clamp(pn, 1e-7, 1 - 1e-7))).sum(dim=1).mean()
    reg = sum(1e-5 * (params[k] ** 2).sum() for k in net.l2_names)
    return p, loss, reg


def test_conv1d_residual_shapes_and_grads_match_torch_autograd():
    from oracle.net import Conv1dResidualNet
    full = Conv1dResidualNet(num_classes=12)
    assert full.L0 == 399
    assert [b['Lout'] for b in full.blocks] == [200, 100] + [100] * 8 + [50, 25, 13]
    assert [(r['Lin'], r['Lout']) for r in full.red] == [(13, 7), (7, 5)] and (full.T, full.C) == (5, 1024)
    # small widths / short clip: same structure incl. odd and even lengths under the strided 3-wide pools
    net = Conv1dResidualNet(num_classes=12, input_size=4000, blocks=[(16, 2), (24, 2), (24, 1), (32, 2)], c0=8,
                            c_reduce=40, dtype=np.float64)
    assert [b['Lin'] for b in net.blocks] == [99, 50, 25, 25]
    rng = np.random.RandomState(8)
    for k in net.params:
        if k.endswith('gamma'):
            net.params[k] = (1.0 + 0.1 * rng.randn(*net.params[k].shape)).astype(np.float32)
        if k.endswith('beta') or k.endswith('bias'):
            net.params[k] = (0.1 * rng.randn(*net.params[k].shape)).astype(np.float32)
    B = 3
    x = (rng.randn(B, 4000) * 0.3).astype(np.float64)
    y = np.eye(12)[[3, 0, 11]]
    loss, p, grads, _ = net.loss_and_grads(x, y, seed=9, step=2)
    tparams = {k: torch.tensor(v.astype(np.float64), requires_grad=True) for k, v in net.params.items()}
    pt, tloss, treg = _torch_residual(net, tparams, x, y, 9, 2)
    (tloss + treg).backward()
    np.testing.assert_allclose(p, pt.detach().numpy(), rtol=1e-9, atol=1e-12)
    assert abs(loss - tloss.item()) < 1e-10
    for k, g in grads.items():
        tg = tparams[k].grad.numpy().reshape(g.shape)
        assert np.abs(g - tg).max() / max(np.abs(tg).max(), 1e-12) < 1e-8, k


# ---- conv_1d_mfcc_and_raw ---------------------------------------------------------------
def test_mfcc_and_raw_shapes_and_grads_match_torch_autograd():
    from oracle.net import MfccAndRawNet
    full = MfccAndRawNet(num_classes=12)
    assert full.L0 == 96 and [b['Lout'] for b in full.blocks] == [96, 96, 48, 48, 24, 24, 12, 12, 6, 6]
    assert (full.T, full.C) == (6, 384)
    assert list(full.params)[:6] == ['conv1d_1/kernel', 'batch_normalization_1/gamma', 'batch_normalization_1/beta',
                                     'conv1d_2/kernel', 'batch_normalization_2/gamma', 'batch_normalization_2/beta']
    assert full.params['conv1d_2/kernel'].shape == (3, 480, 96)
    net = MfccAndRawNet(num_classes=12, spectrogram_length=23, num_features=8, raw_size=480 + 22 * 160,
                        blocks=[(24, 1), (32, 2), (32, 1)], c_mfcc=8, c_raw=16, dtype=np.float64)
    rng = np.random.RandomState(10)
    for k in net.params:
        if k.endswith('gamma'):
            net.params[k] = (1.0 + 0.1 * rng.randn(*net.params[k].shape)).astype(np.float32)
        if k.endswith('beta') or k.endswith('bias'):
            net.params[k] = (0.1 * rng.randn(*net.params[k].shape)).astype(np.float32)
    B = 3
    xm = (rng.randn(B, 23 * 8) * 2.0).astype(np.float64)
    xr = (rng.randn(B, net.L_in) * 0.3).astype(np.float64)
    y = np.eye(12)[[3, 0, 11]]
    loss, p, grads, _ = net.loss_and_grads([xm, xr], y, seed=9, step=2)
    dt = torch.float64
    params = {k: torch.tensor(v.astype(np.float64), requires_grad=True) for k, v in net.params.items()}

    def bn(h, idx, relu=True):
        h = F.batch_norm(h, None, None, params['batch_normalization_%d/gamma' % idx],
                         params['batch_normalization_%d/beta' % idx], training=True, eps=1e-3)
        return torch.clamp(h, 0, 6) if relu else h

    def dwpw(h, dwn, pwn, cin, cout):
        w = params[dwn].reshape(3, cin)
        h = F.conv1d(F.pad(h, (1, 1)), w.t().unsqueeze(1), groups=cin)
        return F.conv1d(h, params[pwn].reshape(cin, cout).t().unsqueeze(2))
    hm = torch.from_numpy(xm).to(dt).reshape(B, 23, 8).permute(0, 2, 1)
    am = bn(F.conv1d(hm, params[net.first_m[0]].permute(2, 1, 0)), net.first_m[1])
    fr = torch.from_numpy(xr).to(dt).unfold(1, 480, 160).permute(0, 2, 1)          # [B, 480, 23]
    ar = bn(F.conv1d(fr, params[net.first_r[0]].permute(2, 1, 0)), net.first_r[1])
    h = torch.cat([am, ar], dim=1)
    for blk in net.blocks:
        if 'short' in blk:
            res = bn(F.conv1d(h, params[blk['short'][0]].reshape(blk['cin'], blk['nf']).t().unsqueeze(2), stride=blk['stride']),
                     blk['short'][1], relu=False)
        else:
            res = h
        a = bn(dwpw(h, blk['dw1'], blk['pw1'], blk['cin'], blk['nf']), blk['bn1'])
        a = bn(dwpw(a, blk['dw2'], blk['pw2'], blk['nf'], blk['nf']), blk['bn2'])
        _, pl, pr = L.same_pad(blk['Lin'], 3, blk['stride'])
        h = F.max_pool1d(F.pad(a, (pl, pr), value=float('-inf')), 3, blk['stride']) + res
    feat = h.mean(dim=2)
    m = torch.from_numpy(L.dropout_mask(L.dropout_key(9, 2, 1), B * net.C, 0.7).reshape(B, net.C)).to(dt)
    pt = torch.softmax((feat * m / 0.7) @ params['dense_1/kernel'] + params['dense_1/bias'], dim=1)
    yt = torch.from_numpy(y).to(dt)
    pn = pt / pt.sum(dim=1, keepdim=True)
    tloss = -(yt * torch.log(torch.clamp(pn, 1e-7, 1 - 1e-7))).sum(dim=1).mean()
    treg = sum(1e-5 * (params[k] ** 2).sum() for k in net.l2_names)
    (tloss + treg).backward()
    np.testing.assert_allclose(p, pt.detach().numpy(), rtol=1e-9, atol=1e-12)
    assert abs(loss - tloss.item()) < 1e-10
    for k, g in grads.items():
        tg = params[k].grad.numpy().reshape(g.shape)
        assert np.abs(g - tg).max() / max(np.abs(tg).max(), 1e-12) < 1e-8, k
