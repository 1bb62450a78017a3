"""Kernel-level parity of the stand-alone classifier-tail entry points (SURVEY 8b minimum set; csrc/ops.hip) against the
oracle: kws_dropout_{fwd,bwd} (bit-exact masks), kws_attn_pool_{fwd,bwd} (model.py:824-827, ties included),
kws_softmax_xent_smooth_{fwd,bwd} (utils.py:87-108, clip edges included), and the RCCL wrapper on a world of one."""
import ctypes

import numpy as np
import pytest
import torch

from oracle import layers as OL
from speech_recognition_amd import _lib

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def S():
    return _lib.stream_ptr()


@pytest.mark.parametrize("B,n,keep,row_offset", [(1, 1, 0.6, 0), (7, 4608, 0.6, 0), (5, 1024, 0.6, 1019), (3, 257, 0.8, 4), (64, 1024, 1.0, 0)])
def test_dropout_masks_are_the_oracles(B, n, keep, row_offset):
    rng = np.random.RandomState(B * 31 + n)
    x = rng.randn(B, n).astype(np.float32)
    out = torch.full((B, n), float("nan"), device="cuda")
    dx_in = dev(x)          # device inputs stay referenced until the synchronising read below (a temporary's block could
                            # be handed to the next allocation while the kernel is still queued)
    _lib.call("kws_dropout_fwd", _lib.ptr(dx_in), _lib.ptr(out), B, n, keep, ctypes.c_uint64(0x1234567890), 17, 2, row_offset, S())
    mask = OL.dropout_mask(OL.dropout_key(0x1234567890, 17, 2), B * n, keep, row_offset * n).reshape(B, n)
    ref = np.where(mask, x * np.float32(1.0 / np.float32(keep)), np.float32(0)).astype(np.float32)
    assert np.array_equal(out.cpu().numpy(), ref)                    # bit-exact: integer hash + one f32 multiply
    dx = torch.empty((B, n), device="cuda")
    _lib.call("kws_dropout_bwd", _lib.ptr(dx_in), _lib.ptr(dx), B, n, keep, ctypes.c_uint64(0x1234567890), 17, 2, row_offset, S())
    assert torch.equal(dx, out)
    if keep < 1.0 and B * n > 1000:
        assert abs(mask.mean() - keep) < 0.03
    lib = _lib.load()
    assert lib.kws_dropout_fwd(_lib.ptr(out), _lib.ptr(out), B, n, 0.0, ctypes.c_uint64(1), 0, 1, 0, S()) != 0   # keep_prob 0


@pytest.mark.parametrize("B,T,C", [(1, 9, 512), (6, 9, 512), (3, 5, 300), (2, 1, 64), (4, 12, 33)])
def test_attn_pool_fwd_bwd_matches_oracle(B, T, C):
    rng = np.random.RandomState(B + T + C)
    x = np.clip(rng.randn(B, T, C) * 2.0 + 1.0, 0, 6).astype(np.float32)          # post-ReLU6 activations: many exact 0s / 6s
    att = OL.softmax(rng.randn(B, T), axis=1).astype(np.float32)
    if T > 2:                                                                      # exact ties between time steps
        att[:, 1] = att[:, 0]
        x[:, 1, ::3] = x[:, 0, ::3]
    feat = torch.full((B, 2 * C), float("nan"), device="cuda")
    d_x, d_att = dev(x), dev(att)
    _lib.call("kws_attn_pool_fwd", _lib.ptr(d_x), _lib.ptr(d_att), _lib.ptr(feat), B, T, C, S())
    xa = x * att[:, :, None]                                                      # f32 product, as on the device
    ref = np.concatenate([xa.max(axis=1), x.astype(np.float64).mean(axis=1)], axis=1)
    got = feat.cpu().numpy()
    assert np.array_equal(got[:, :C], xa.max(axis=1))                              # max of f32 products: bit-exact
    np.testing.assert_allclose(got[:, C:], ref[:, C:], rtol=2e-6, atol=1e-7)
    dfeat = rng.randn(B, 2 * C).astype(np.float32)
    d_dfeat = dev(dfeat)
    lib = _lib.load()
    ws = torch.empty(int(lib.kws_attn_pool_bwd_workspace_floats(B, T, C)), device="cuda")
    dx = torch.full((B, T, C), float("nan"), device="cuda")
    datt = torch.full((B, T), float("nan"), device="cuda")
    _lib.call("kws_attn_pool_bwd", _lib.ptr(d_x), _lib.ptr(d_att), _lib.ptr(d_dfeat), _lib.ptr(dx), _lib.ptr(datt),
              _lib.ptr(ws), B, T, C, S())
    # oracle (oracle/net.py:loss_and_grads tail): ties share the max gradient equally
    ind = (xa == xa.max(axis=1, keepdims=True)).astype(np.float64)
    ind = ind / ind.sum(axis=1, keepdims=True)
    dxa = ind * dfeat[:, None, :C].astype(np.float64)
    ref_dx = dxa * att[:, :, None] + dfeat[:, None, C:].astype(np.float64) / T
    ref_datt = (dxa * x).sum(axis=2)
    np.testing.assert_allclose(dx.cpu().numpy(), ref_dx, rtol=2e-6, atol=1e-7)
    np.testing.assert_allclose(datt.cpu().numpy(), ref_datt, rtol=1e-5, atol=1e-5)
    if T > 2 and C >= 64:
        assert (ind.max(axis=1) < 1).any()                                         # the tie path was exercised


@pytest.mark.parametrize("B,NC,s", [(1, 12, 0.1), (37, 12, 0.1), (5, 32, 0.1), (9, 12, 0.0), (300, 12, 0.1)])
def test_softmax_xent_smooth_fwd_bwd_matches_oracle(B, NC, s):
    rng = np.random.RandomState(B * 7 + NC)
    p = OL.softmax(rng.randn(B, NC) * 3.0, axis=1).astype(np.float32)
    p[0] = 0.0
    p[0, 3] = 1.0                                                                  # both clip edges active
    lab = rng.randint(0, NC, B)
    y = np.eye(NC, dtype=np.float32)[lab]
    per = torch.full((B,), float("nan"), device="cuda")
    cor = torch.full((B,), float("nan"), device="cuda")
    d_p, d_y = dev(p), dev(y)
    _lib.call("kws_softmax_xent_smooth_fwd", _lib.ptr(d_p), _lib.ptr(d_y), _lib.ptr(per), _lib.ptr(cor), B, NC, s, S())
    loss, per_ref, dp_ref = OL.smooth_cce_fwd_bwd(p.astype(np.float64), y.astype(np.float64), s)
    np.testing.assert_allclose(per.cpu().numpy(), per_ref, rtol=2e-6, atol=2e-6)
    assert np.array_equal(cor.cpu().numpy(), (p.argmax(1) == lab).astype(np.float32))
    dp = torch.full((B, NC), float("nan"), device="cuda")
    dl = torch.full((B, NC), float("nan"), device="cuda")
    inv = 1.0 / (2 * B)                                                            # a data-parallel world of 2
    _lib.call("kws_softmax_xent_smooth_bwd", _lib.ptr(d_p), _lib.ptr(d_y), _lib.ptr(dp), _lib.ptr(dl), B, NC, s, inv, S())
    ref_dp = dp_ref * 0.5                                                          # oracle scales by 1/B
    scale = np.abs(ref_dp).max()
    assert np.abs(dp.cpu().numpy() - ref_dp).max() <= 2e-6 * scale               # (B = 1: the only row is the clipped one)
    ref_dl = OL.softmax_bwd(ref_dp, p.astype(np.float64), axis=1)
    assert np.abs(dl.cpu().numpy() - ref_dl).max() <= 2e-6 * max(np.abs(ref_dl).max(), 1e-12)
    assert (dp[0].cpu().numpy() == 0).all()                                        # clipped entries pass no gradient
    # either output may be omitted
    _lib.call("kws_softmax_xent_smooth_bwd", _lib.ptr(d_p), _lib.ptr(d_y), None, _lib.ptr(dl), B, NC, s, inv, S())
    lib = _lib.load()
    assert lib.kws_softmax_xent_smooth_bwd(_lib.ptr(d_p), _lib.ptr(d_y), None, None, B, NC, s, inv, S()) != 0
    assert lib.kws_softmax_xent_smooth_fwd(_lib.ptr(d_p), _lib.ptr(d_y), _lib.ptr(per), None, B, 65, s, S()) != 0


def test_standalone_ops_reproduce_the_fused_tail():
    """The fused tail of the network program and the stand-alone entry points are the same arithmetic: feed the
    stand-alone chain the program's own last activation and attention weights and compare feat -> loss with the
    program's probabilities / loss (inference-free check of the composition the header documents)."""
    from oracle.net import TimeSlicedAttentionNet
    from speech_recognition_amd.net import DeviceNet
    B = 5
    ora = TimeSlicedAttentionNet(num_classes=12, dtype=np.float64)
    net = DeviceNet(_lib.KWS_NET_TS_ATTENTION, 12)
    net.set_weights(dict(ora.params, **ora.state))
    rng = np.random.RandomState(2)
    x = (rng.randn(B, 16000) * 0.0774).astype(np.float32)
    lab = rng.randint(0, 12, B)
    y = np.eye(12, dtype=np.float32)[lab]
    d_x0, d_y0 = dev(x), dev(y)
    probs = net.train_fwd_bwd(d_x0, d_y0, seed=5, step=1).cpu().numpy()
    torch.cuda.synchronize()
    y12 = net.debug_view(B, 0, 11).reshape(B, 9, 512)
    bn = net.debug_view(B, 2, 11)
    a = np.clip((y12.astype(np.float64) * bn[:512] + bn[512:1024]).astype(np.float32), 0, 6)
    att = net.debug_view(B, 3, 0).reshape(B, 9)
    feat = torch.empty((B, 1024), device="cuda")
    d_a, d_att = dev(a), dev(att)
    _lib.call("kws_attn_pool_fwd", _lib.ptr(d_a), _lib.ptr(d_att), _lib.ptr(feat), B, 9, 512, S())
    fd = torch.empty_like(feat)
    _lib.call("kws_dropout_fwd", _lib.ptr(feat), _lib.ptr(fd), B, 1024, 0.6, ctypes.c_uint64(5), 1, 2, 0, S())
    W2 = ora.params['dense_2/kernel'].astype(np.float64)
    p = OL.softmax(fd.cpu().numpy().astype(np.float64) @ W2, axis=1)
    assert np.abs(p - probs).max() < 2e-6
    per = torch.empty(B, device="cuda")
    d_probs, d_y = dev(probs), dev(y)
    _lib.call("kws_softmax_xent_smooth_fwd", _lib.ptr(d_probs), _lib.ptr(d_y), _lib.ptr(per), None, B, 12, 0.1, S())
    assert abs(float(per.sum().item()) - float(net.metrics[0].item())) < 1e-5 * B


def test_rccl_wrapper_world_of_one():
    """kws_comm_* / kws_allreduce_grads on a single rank: the communicator comes up from the (rank, world, id) triple
    and the in-place sum over one rank leaves the buffer unchanged.  (N > 1 ranks need N GPUs: SCALE runs.)"""
    lib = _lib.load()
    uid = ctypes.create_string_buffer(128)
    _lib.check(lib.kws_comm_unique_id(uid), "kws_comm_unique_id")
    comm = ctypes.c_void_p()
    _lib.check(lib.kws_comm_create(0, 1, uid, ctypes.byref(comm)), "kws_comm_create")
    g = torch.arange(1191433, dtype=torch.float32, device="cuda") * 1e-3
    ref = g.clone()
    _lib.call("kws_allreduce_grads", comm, _lib.ptr(g), g.numel(), S())
    torch.cuda.synchronize()
    assert torch.equal(g, ref)
    _lib.check(lib.kws_comm_destroy(comm), "kws_comm_destroy")
    assert lib.kws_comm_create(3, 2, uid, ctypes.byref(comm)) != 0          # rank outside the world
