"""The bit-identity test of round 5's experiment (weight-gradient work items beside the depthwise-backward passes).  It
needs a library built with mode3.patch applied (see README.md here); it is not collected by the repo's test suite."""
import ctypes

import numpy as np
import pytest
import torch

from speech_recognition_amd import _lib
from tests.test_kernels_gpu import S, dev, rel_err

pytestmark = pytest.mark.gpu


# Round 5: a depthwise-backward pass and weight-gradient work of the same layer in ONE grid (kws_dwconv_bwd_bn_wgrad_f32), the
# weight-gradient GEMM cut by item range AND by stage window with parked accumulators.  Shapes: all four tile forms of the
# wave-specialised kernel (K, N multiples of 128 or only of 64), both strides, a batch whose last split is ragged.
@pytest.mark.parametrize("Lin,stride,pad_l,C,N,B", [(99, 1, 0, 256, 256, 40), (97, 2, 1, 256, 320, 40), (49, 1, 0, 320, 384, 70),
                                                    (47, 2, 1, 320, 320, 70), (399, 1, 0, 128, 128, 12)])
def test_wgrad_items_beside_depthwise_passes_are_bit_identical(Lin, stride, pad_l, C, N, B):
    lib = _lib.load()
    Lout = Lin - 2 if stride == 1 else (Lin + 1) // 2
    K, M = C, B * Lout
    rng = np.random.RandomState(Lin + C)
    y = dev(rng.randn(B, Lin, C).astype(np.float32) * 2.0)
    w = dev(rng.randn(3, C).astype(np.float32))
    gamma = (1 + 0.1 * rng.randn(C)).astype(np.float32)
    mean, rstd = rng.randn(C).astype(np.float32) * 0.1, (1 + 0.1 * rng.rand(C)).astype(np.float32)
    bn = dev(np.concatenate([gamma * rstd, 0.5 - mean * gamma * rstd, mean, rstd]).astype(np.float32))
    dz = dev(rng.randn(B, Lout, C).astype(np.float32))
    coef = dev(rng.randn(2 * C).astype(np.float32) * 0.1)
    z = rng.randn(M, K).astype(np.float32)
    dY = rng.randn(M, N).astype(np.float32)
    dz_, dY_ = dev(z), dev(dY)
    n_part = int(lib.kws_dwconv_bwd_part_floats(B, Lin, C))
    gran = ctypes.c_int(0)
    items = int(lib.kws_gemm_tn_items(M, K, N, ctypes.byref(gran)))
    assert items > 0 and items % 8 == 0 and items % gran.value == 0
    wsf = int(lib.kws_gemm_tn_workspace_floats(M, K, N))
    ck = torch.full((int(lib.kws_gemm_tn_ckpt_floats(M, K, N)),), float("nan"), device="cuda")
    Sout = ctypes.c_int(0)

    def items_of(ws, lo, hi, f0=0, f1=1024, resume=()):
        wi = _lib.WgradItems()
        wi.Z = dz_.data_ptr(); wi.dY = dY_.data_ptr(); wi.M = M; wi.K = K; wi.N = N; wi.slabs = ws.data_ptr(); wi.ckpt = ck.data_ptr()
        wi.item_lo = lo; wi.item_hi = hi; wi.f0 = f0; wi.f1 = f1; wi.n_resume = len(resume)
        for i, (rl, rh, rf) in enumerate(resume):
            wi.resume_lo[i] = rl; wi.resume_hi[i] = rh; wi.resume_f[i] = rf
        return wi

    def run(pas, G, wi, part=None, dy=None):
        _lib.call("kws_dwconv_bwd_bn_wgrad_f32", _lib.ptr(dz), _lib.ptr(y), _lib.ptr(bn), _lib.ptr(w), _lib.ptr(coef) if pas == 2 else None,
                  _lib.ptr(dy), _lib.ptr(part), pas, B, Lin, Lout, C, stride, pad_l, ctypes.byref(wi), G, ctypes.byref(Sout), S())

    # the reference: the separate calls
    part0 = torch.full((n_part,), float("nan"), device="cuda")
    dy0 = torch.full((B, Lin, C), float("nan"), device="cuda")
    _lib.call("kws_dwconv_bwd_bn_f32", _lib.ptr(dz), _lib.ptr(y), _lib.ptr(bn), _lib.ptr(w), None, None, _lib.ptr(part0), 1, B, Lin, Lout, C, stride, pad_l, S())
    _lib.call("kws_dwconv_bwd_bn_f32", _lib.ptr(dz), _lib.ptr(y), _lib.ptr(bn), _lib.ptr(w), _lib.ptr(coef), _lib.ptr(dy0), None, 2, B, Lin, Lout, C, stride, pad_l, S())
    dW0 = torch.full((K, N), float("nan"), device="cuda")
    ws0 = torch.empty(wsf, device="cuda")
    _lib.call("kws_gemm_tn_f32", _lib.ptr(dz_), _lib.ptr(dY_), _lib.ptr(dW0), M, K, N, _lib.ptr(ws0), S())
    assert rel_err(dW0.cpu().numpy(), z.astype(np.float64).T @ dY.astype(np.float64)) < 5e-6
    # (a) items alone (pass 0) = the weight-gradient kernel's slabs
    ws1 = torch.full((wsf,), float("nan"), device="cuda")
    run(0, 0, items_of(ws1, 0, items))
    used = Sout.value * K * N
    torch.cuda.synchronize()
    assert torch.equal(ws1[:used], ws0[:used])
    # (b) cut by item range and stage window: pass 1 beside the first third of the items up to 37 %, pass 2 beside the second third
    # up to 61 % and then on to 80 % in a launch of its own, the remainder (and the untouched last third) at the end
    a, b = items // 3 // 8 * 8, 2 * items // 3 // 8 * 8
    for G in (8, 64, 256):
        ws2 = torch.full((wsf,), float("nan"), device="cuda")
        part2 = torch.full((n_part,), float("nan"), device="cuda")
        dy2 = torch.full((B, Lin, C), float("nan"), device="cuda")
        ck.fill_(float("nan"))
        run(1, G, items_of(ws2, 0, a, 0, 379), part=part2)
        run(2, G, items_of(ws2, a, b, 0, 625), dy=dy2)
        run(0, 0, items_of(ws2, a, b, 625, 820))
        run(0, 0, items_of(ws2, 0, items, 0, 1024, resume=((0, a, 379), (a, b, 820))))
        torch.cuda.synchronize()
        assert torch.equal(part2, part0), G
        assert torch.equal(dy2, dy0), G
        assert torch.equal(ws2[:used], ws0[:used]), G
    # (c) argument checks: nothing is launched for a bad description
    for bad in (dict(lo=4, hi=items), dict(lo=0, hi=items + 8), dict(lo=0, hi=items, f0=600, f1=500), dict(lo=0, hi=items, f1=1025)):
        wi = items_of(ws1, bad["lo"], bad["hi"], bad.get("f0", 0), bad.get("f1", 1024))
        with pytest.raises(_lib.KwsError):
            run(0, 0, wi)
    wi = items_of(ws1, 0, items, 0, 512)
    wi.ckpt = None
    with pytest.raises(_lib.KwsError):          # a cut window without the checkpoint buffer
        run(0, 0, wi)
    with pytest.raises(_lib.KwsError):          # workgroups of the pass: a multiple of 8, at most 256
        run(1, 12, items_of(ws1, 0, 0), part=part0)
