"""The fp16 x 2 split GEMMs (A/B arm, csrc/gemm_f16x2.hip): every f32 operand scaled by a power of two taken from
its |x| maximum, split into two fp16 parts, three f16 MFMA products accumulated in f32.  As accurate against float64 as
the f32-MFMA kernels the product uses, over the magnitudes a training step produces (activations of order one,
gradients of order 1e-7) and beyond.  Off by default (kws_net_set_gemm_mode(net, 2))."""
import ctypes

import numpy as np
import pytest
import torch

from speech_recognition_amd import _lib

pytestmark = pytest.mark.gpu

WORDS = 256            # KWS_ABSMAX_WORDS: one slot group


def absmax_slots(*tensors):
    """slot groups (|x| maxima) of the tensors through kws_absmax_batch_f32"""
    n = len(tensors)
    slots = torch.full((n, WORDS), 0x7FFFFFFF, dtype=torch.int32, device="cuda")      # the call zeroes them itself
    P, L = ctypes.c_void_p * n, ctypes.c_int64 * n
    _lib.call("kws_absmax_batch_f32", P(*[t.data_ptr() for t in tensors]), L(*[t.numel() for t in tensors]),
              _lib.ptr(slots), n, _lib.stream_ptr())
    return slots


def test_absmax_slots_hold_the_exact_maximum():
    g = torch.Generator(device="cuda")
    g.manual_seed(3)
    a = torch.randn(1000003, generator=g, device="cuda") * 3.0
    a[77777] = -123.456
    b = torch.zeros(10, device="cuda")
    c = torch.randn(5, generator=g, device="cuda") * 1e-30
    s = absmax_slots(a, b, c)
    got = s.view(3, 16, 16)[:, :, 0].contiguous().view(torch.float32).max(dim=1).values
    assert float(got[0]) == float(a.abs().max()) == float(np.float32(123.456))
    assert float(got[1]) == 0.0
    assert float(got[2]) == float(c.abs().max())
    untouched = s.view(3, 16, 16)[:, :, 1:]
    assert int(untouched.abs().max()) == 0


@pytest.mark.parametrize("M,K,N,a_scale,w_scale", [
    (1, 128, 128, 1.7, 0.1), (130, 128, 128, 1.7, 0.1), (4096, 192, 192, 1e-7, 0.1), (3000, 320, 320, 3e4, 1e-3),
    (9216, 512, 512, 1.0, 0.05), (50000, 256, 320, 2e-9, 30.0), (777, 384, 512, 1e12, 1e-15), (515, 128, 192, 0.0, 0.1)])
def test_f16x2_forward_matches_float64_as_well_as_the_f32_kernel(M, K, N, a_scale, w_scale):
    g = torch.Generator(device="cuda")
    g.manual_seed(M + K)
    A = torch.randn((M, K), generator=g, device="cuda") * a_scale
    if a_scale == 1.7:
        A[0, :4] = torch.tensor([1e-30, -3e4, 65504.0, 1.0 + 2.0 ** -20], device="cuda")   # tiny / large / many mantissa bits
    A *= torch.exp(torch.randn((M, 1), generator=g, device="cuda") * 2.0)                   # rows of very different size
    W = torch.randn((K, N), generator=g, device="cuda") * w_scale
    slots = absmax_slots(A, W)
    Wp = torch.empty((2, N, K), dtype=torch.float16, device="cuda")
    P, I = ctypes.c_void_p * 1, ctypes.c_int * 1
    _lib.call("kws_f16x2_split_batch", P(W.data_ptr()), P(Wp.data_ptr()), I(K), I(N), I(1), P(slots[1].data_ptr()), 1,
              _lib.stream_ptr())
    assert torch.isfinite(Wp.float()).all() and float(Wp[0].float().abs().max()) < 65504.0
    C2 = torch.full((M, N), float("nan"), device="cuda")
    stats = torch.full((_lib.load().kws_gemm_nn_f16x2_stats_rows(M), 2, N), float("nan"), device="cuda")
    _lib.call("kws_gemm_nn_f16x2_f32", _lib.ptr(A), _lib.ptr(Wp), _lib.ptr(C2), M, K, N, _lib.ptr(slots[0]), _lib.ptr(slots[1]),
              _lib.ptr(stats), _lib.stream_ptr())
    ref = A.double() @ W.double()
    assert torch.isfinite(C2).all()
    if a_scale == 0.0:
        assert float(C2.abs().max()) == 0.0
        return
    scale = float(ref.abs().max())
    e2 = float((C2.double() - ref).abs().max()) / scale
    assert float((stats[:, 0].double().sum(0) - C2.double().sum(0)).abs().max()) < 1e-3 * max(scale, float(C2.double().sum(0).abs().max()))
    C2b = torch.empty((M, N), device="cuda")        # without the statistics epilogue: the same product, bit for bit
    _lib.call("kws_gemm_nn_f16x2_f32", _lib.ptr(A), _lib.ptr(Wp), _lib.ptr(C2b), M, K, N, _lib.ptr(slots[0]), _lib.ptr(slots[1]),
              None, _lib.stream_ptr())
    assert torch.equal(C2, C2b)
    C1 = torch.empty((M, N), device="cuda")
    _lib.call("kws_gemm_nn_f32", _lib.ptr(A), _lib.ptr(W), _lib.ptr(C1), M, K, N, None, _lib.stream_ptr())
    e1 = float((C1.double() - ref).abs().max()) / scale
    print("M=%d K=%d N=%d |A|~%.0e: f16x2 %.2e, f32 MFMA %.2e of the maximum" % (M, K, N, a_scale, e2, e1))
    assert e2 < max(2.0 * e1, 4e-7)
    assert e2 < 1e-6                 # the f32 kernels' own test bar (tests/test_kernels_gpu.py)
    # rows far smaller than the tensor's maximum keep their RELATIVE accuracy too (per-row error against the row's own size)
    row_ref = ref.abs().max(dim=1).values
    row_err = (C2.double() - ref).abs().max(dim=1).values
    big = row_ref > scale * 2.0 ** -12
    assert float((row_err[big] / row_ref[big]).max()) < 4e-6


@pytest.mark.parametrize("M,K,N,z_scale,g_scale", [
    (64, 128, 128, 1.3, 0.2), (100, 64, 64, 1.3, 1e-7), (4099, 192, 192, 2.0, 3e-8), (3000, 320, 384, 1e3, 1e-3),
    (9216, 512, 512, 1.0, 1e-6), (50001, 256, 320, 1.3, 2e-7), (777, 384, 128, 1e-20, 1e10), (33, 128, 192, 1.0, 0.0)])
def test_f16x2_weight_gradient_matches_float64_as_well_as_the_f32_kernel(M, K, N, z_scale, g_scale):
    g = torch.Generator(device="cuda")
    g.manual_seed(M + N)
    Z = torch.randn((M, K), generator=g, device="cuda") * z_scale
    G = torch.randn((M, N), generator=g, device="cuda") * g_scale
    lib = _lib.load()
    slots = absmax_slots(Z, G)
    ws2 = torch.empty(lib.kws_gemm_tn_f16x2_workspace_floats(M, K, N), device="cuda")
    D2 = torch.full((K, N), float("nan"), device="cuda")
    _lib.call("kws_gemm_tn_f16x2_f32", _lib.ptr(Z), _lib.ptr(G), _lib.ptr(D2), M, K, N, _lib.ptr(slots[0]), _lib.ptr(slots[1]),
              _lib.ptr(ws2), _lib.stream_ptr())
    ref = Z.double().t() @ G.double()
    assert torch.isfinite(D2).all()
    if g_scale == 0.0:
        assert float(D2.abs().max()) == 0.0
        return
    scale = float(ref.abs().max())
    e2 = float((D2.double() - ref).abs().max()) / scale
    ws1 = torch.empty(lib.kws_gemm_tn_workspace_floats(M, K, N), device="cuda")
    D1 = torch.empty((K, N), device="cuda")
    _lib.call("kws_gemm_tn_f32", _lib.ptr(Z), _lib.ptr(G), _lib.ptr(D1), M, K, N, _lib.ptr(ws1), _lib.stream_ptr())
    e1 = float((D1.double() - ref).abs().max()) / scale
    print("M=%d K=%d N=%d |G|~%.0e: f16x2 %.2e, f32 MFMA %.2e of the maximum" % (M, K, N, g_scale, e2, e1))
    assert e2 < max(2.0 * e1, 4e-7)
    assert e2 < 2e-6
    D2b = torch.empty((K, N), device="cuda")        # bit-reproducible: fixed split and summation order
    _lib.call("kws_gemm_tn_f16x2_f32", _lib.ptr(Z), _lib.ptr(G), _lib.ptr(D2b), M, K, N, _lib.ptr(slots[0]), _lib.ptr(slots[1]),
              _lib.ptr(ws2), _lib.stream_ptr())
    assert torch.equal(D2, D2b)


def _slot_max(slots):
    return float(slots.view(16, 16)[:, 0].contiguous().view(torch.float32).max())


@pytest.mark.parametrize("B,Lin,C,stride", [(3, 37, 128, 1), (5, 22, 192, 2), (2, 99, 320, 1), (7, 11, 512, 1), (1, 5, 64, 2)])
def test_producers_leave_the_exact_maximum_of_the_operand_they_write(B, Lin, C, stride):
    """dwconv_fwd (z), pass 2 of dwconv_bwd and bn_bwd_apply (dy) commit max|x| of what they store - incl. the channel counts
    whose workgroups end in a partial wave (192: 240 threads) - through the library's internal entry points"""
    import ctypes as C_
    lib = C_.CDLL(_lib.LIB_PATH)
    g = torch.Generator(device="cuda")
    g.manual_seed(B * 1000 + C)
    pad_l = 0 if stride == 1 else Lin % 2              # TF 'same', right-biased: (0, 1) for even, (1, 1) for odd lengths
    Lout = Lin - 2 if stride == 1 else (Lin + 1) // 2
    y = torch.randn((B, Lin, C), generator=g, device="cuda") * 2.0
    bn = torch.cat([torch.rand(C, generator=g, device="cuda") + 0.5, torch.randn(C, generator=g, device="cuda") * 0.3,
                    torch.randn(C, generator=g, device="cuda") * 0.1, torch.rand(C, generator=g, device="cuda") + 0.5]).contiguous()
    w = torch.randn((3, C), generator=g, device="cuda")
    S = _lib.stream_ptr()
    P = C_.c_void_p
    # forward
    z = torch.empty((B, Lout, C), device="cuda")
    slots = torch.zeros(256, dtype=torch.int32, device="cuda")
    rc = lib.kws_dwconv_fwd_amax_f32(P(y.data_ptr()), P(bn.data_ptr()), P(w.data_ptr()), P(z.data_ptr()), B, Lin, Lout, C, stride,
                                     pad_l, P(slots.data_ptr()), S)
    _lib.check(rc, 'internal entry point')
    assert _slot_max(slots) == float(z.abs().max()) > 0.0
    z_ref = torch.empty_like(z)                      # the same kernel without the commit writes the same z
    _lib.call("kws_dwconv_fwd_f32", _lib.ptr(y), _lib.ptr(bn), _lib.ptr(w), _lib.ptr(z_ref), B, Lin, Lout, C, stride, pad_l, S)
    assert torch.equal(z, z_ref)
    # backward pass 2
    dz = torch.randn((B, Lout, C), generator=g, device="cuda") * 1e-6
    coef = torch.randn(2 * C, generator=g, device="cuda") * 1e-7
    dy = torch.empty((B, Lin, C), device="cuda")
    slots.zero_()
    rc = lib.kws_dwconv_bwd_bn_amax_f32(P(dz.data_ptr()), P(y.data_ptr()), P(bn.data_ptr()), P(w.data_ptr()), P(coef.data_ptr()),
                                        P(dy.data_ptr()), None, 2, B, Lin, Lout, C, stride, pad_l, P(slots.data_ptr()), S)
    _lib.check(rc, 'internal entry point')
    assert _slot_max(slots) == float(dy.abs().max()) > 0.0
    # BatchNorm backward of the last block
    gdy = torch.randn((B * Lin, C), generator=g, device="cuda") * 1e-5
    gamma = torch.rand(C, generator=g, device="cuda") + 0.5
    slots.zero_()
    rc = lib.kws_bn_bwd_apply_amax(P(gdy.data_ptr()), P(y.data_ptr()), P(bn.data_ptr()), P(gamma.data_ptr()), P(coef.data_ptr()),
                                   C_.c_int64(B * Lin), C, P(slots.data_ptr()), S)
    _lib.check(rc, 'internal entry point')
    assert _slot_max(slots) == float(gdy.abs().max()) > 0.0
