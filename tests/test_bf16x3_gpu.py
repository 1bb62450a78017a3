"""The bf16 x 3 split GEMMs (EXPERIMENT, csrc/gemm_bf16x3.hip): as accurate against float64 as the
f32-MFMA kernels the product uses - which is what would let them replace those kernels without moving a parity
tolerance.  Off by default (KWS_GEMM_BF16X3 / kws_net_set_gemm_mode)."""
import ctypes

import numpy as np
import pytest
import torch

from speech_recognition_amd import _lib

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("M,K,N", [(1, 128, 128), (130, 128, 128), (4096, 192, 192), (3000, 320, 320), (9216, 512, 512),
                                   (50000, 256, 320), (777, 384, 512)])
def test_bf16x3_matches_float64_as_well_as_the_f32_kernel(M, K, N):
    g = torch.Generator(device="cuda")
    g.manual_seed(M + K)
    A = torch.randn((M, K), generator=g, device="cuda") * 1.7
    A[0, :4] = torch.tensor([1e-30, -3e4, 65504.0, 1.0 + 2.0 ** -20], device="cuda")     # tiny / large / many mantissa bits
    W = torch.randn((K, N), generator=g, device="cuda") * 0.1
    Wt = W.t().contiguous()
    Wp = torch.empty((3, N, K), dtype=torch.bfloat16, device="cuda")       # the kernel split once: planes of [N][K]
    P, I = ctypes.c_void_p * 1, ctypes.c_int * 1
    _lib.call("kws_bf16x3_split_batch", P(W.data_ptr()), P(Wp.data_ptr()), I(K), I(N), I(1), 1, _lib.stream_ptr())
    assert float((Wp.double().sum(0) - Wt.double()).abs().max()) <= 2.0 ** -22 * float(Wt.abs().max())   # hi + mid + lo = x
    C3 = torch.full((M, N), float("nan"), device="cuda")
    stats = torch.full((_lib.load().kws_gemm_nn_bf16x3_stats_rows(M), 2, N), float("nan"), device="cuda")
    _lib.call("kws_gemm_nn_bf16x3p_f32", _lib.ptr(A), _lib.ptr(Wp), _lib.ptr(C3), M, K, N, _lib.ptr(stats), _lib.stream_ptr())
    ref = A.double() @ W.double()
    scale = float(ref.abs().max())
    e3 = float((C3.double() - ref).abs().max()) / scale
    assert torch.isfinite(C3).all()
    assert float((stats[:, 0].double().sum(0) - C3.double().sum(0)).abs().max()) < 1e-3 * max(1.0, float(C3.double().sum(0).abs().max()))
    assert float((stats[:, 1].double().sum(0) - (C3.double() ** 2).sum(0)).abs().max()) < 1e-4 * float((C3.double() ** 2).sum(0).max())
    C3b = torch.empty((M, N), device="cuda")         # without the statistics epilogue: the same product, bit for bit
    _lib.call("kws_gemm_nn_bf16x3p_f32", _lib.ptr(A), _lib.ptr(Wp), _lib.ptr(C3b), M, K, N, None, _lib.stream_ptr())
    assert torch.equal(C3, C3b)
    if K % 64 == 0 and N % 64 == 0:
        C1 = torch.empty((M, N), device="cuda")
        _lib.call("kws_gemm_nn_f32", _lib.ptr(A), _lib.ptr(W), _lib.ptr(C1), M, K, N, None, _lib.stream_ptr())
        e1 = float((C1.double() - ref).abs().max()) / scale
        print("M=%d K=%d N=%d: bf16x3 %.2e, f32 MFMA %.2e of the maximum" % (M, K, N, e3, e1))
        assert e3 < max(2.0 * e1, 4e-7)
    assert e3 < 1e-6                 # the f32 kernels' own test bar (tests/test_kernels_gpu.py)


@pytest.mark.parametrize("M,K,N", [(64, 128, 128), (100, 64, 64), (4099, 192, 192), (3000, 320, 384), (9216, 512, 512),
                                   (50001, 256, 320), (777, 384, 128), (33, 128, 192)])
def test_bf16x3_weight_gradient_matches_float64_as_well_as_the_f32_kernel(M, K, N):
    g = torch.Generator(device="cuda")
    g.manual_seed(M + N)
    Z = torch.randn((M, K), generator=g, device="cuda") * 1.3
    G = torch.randn((M, N), generator=g, device="cuda") * 0.2
    Z[0, :3] = torch.tensor([1e-30, -3e4, 1.0 + 2.0 ** -20], device="cuda")
    lib = _lib.load()
    ws3 = torch.empty(lib.kws_gemm_tn_bf16x3_workspace_floats(M, K, N), device="cuda")
    D3 = torch.full((K, N), float("nan"), device="cuda")
    _lib.call("kws_gemm_tn_bf16x3_f32", _lib.ptr(Z), _lib.ptr(G), _lib.ptr(D3), M, K, N, _lib.ptr(ws3), _lib.stream_ptr())
    ref = Z.double().t() @ G.double()
    scale = float(ref.abs().max())
    e3 = float((D3.double() - ref).abs().max()) / scale
    assert torch.isfinite(D3).all()
    ws1 = torch.empty(lib.kws_gemm_tn_workspace_floats(M, K, N), device="cuda")
    D1 = torch.empty((K, N), device="cuda")
    _lib.call("kws_gemm_tn_f32", _lib.ptr(Z), _lib.ptr(G), _lib.ptr(D1), M, K, N, _lib.ptr(ws1), _lib.stream_ptr())
    e1 = float((D1.double() - ref).abs().max()) / scale
    print("M=%d K=%d N=%d: bf16x3 %.2e, f32 MFMA %.2e of the maximum" % (M, K, N, e3, e1))
    assert e3 < max(2.0 * e1, 4e-7)
    assert e3 < 2e-6
    D3b = torch.empty((K, N), device="cuda")        # bit-reproducible: fixed split and summation order
    _lib.call("kws_gemm_tn_bf16x3_f32", _lib.ptr(Z), _lib.ptr(G), _lib.ptr(D3b), M, K, N, _lib.ptr(ws3), _lib.stream_ptr())
    assert torch.equal(D3, D3b)
