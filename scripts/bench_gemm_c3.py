"""Per-layer GEMM micro-benchmark on the pointwise shapes of BASELINE configs[2] (C3: conv_1d_log_mfcc, batch 2048, 96 frames
after the first convolution; reference model.py:1400-1479): forward (with BN statistics), input gradient and weight gradient as
separate launches, HIP events on the launch stream, next to each shape's two floors (HBM at the measured 6.3 TB/s copy rate and
the f32 MFMA peak 157.3 TFLOP/s).  KWS_LIB_PATH selects a variant build (scripts/build_variant.sh)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from speech_recognition_amd import _lib  # noqa: E402

lib = _lib.load()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
# (rows per clip, K, N, how many times the shape occurs in the net)
SHAPES = [(96, 64, 64, 4), (96, 64, 128, 1), (96, 128, 128, 1), (48, 128, 128, 2), (48, 128, 192, 1), (48, 192, 192, 1),
          (24, 192, 192, 4), (24, 192, 256, 1), (24, 256, 256, 1), (12, 256, 256, 4)]
S = _lib.stream_ptr()


def timeit(fn, n=20):
    fn()
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


tot = [0.0, 0.0, 0.0, 0.0]
print("lib: %s" % _lib.LIB_PATH)
for L, K, N, cnt in SHAPES:
    M = B * L
    A = torch.randn(M, K, device='cuda')
    W = torch.randn(K, N, device='cuda') * 0.1
    C = torch.empty(M, N, device='cuda')
    G = torch.randn(M, N, device='cuda')
    WT = W.t().contiguous()
    DZ = torch.empty(M, K, device='cuda')
    dW = torch.empty(K, N, device='cuda')
    part = torch.empty(lib.kws_gemm_num_row_tiles(M) * 2 * N, device='cuda')
    ws = torch.empty(int(lib.kws_gemm_tn_workspace_floats(M, K, N)), device='cuda')
    t1 = timeit(lambda: _lib.call("kws_gemm_nn_f32", _lib.ptr(A), _lib.ptr(W), _lib.ptr(C), M, K, N, _lib.ptr(part), S))
    t2 = timeit(lambda: _lib.call("kws_gemm_nn_f32", _lib.ptr(G), _lib.ptr(WT), _lib.ptr(DZ), M, N, K, None, S))
    t3 = timeit(lambda: _lib.call("kws_gemm_tn_f32", _lib.ptr(A), _lib.ptr(G), _lib.ptr(dW), M, K, N, _lib.ptr(ws), S))
    f = 2.0 * M * K * N
    by = 4.0 * M * (K + N)
    hbm_us, mfma_us = by / 6.3e12 * 1e6, f / 157.3e12 * 1e6
    print("M=%7d K=%3d N=%3d x%d | fwd %6.1f us %5.1f TF %4.2f TB/s | dgrad %6.1f us | wgrad %6.1f us | floors: hbm %5.1f us, mfma %5.1f us, "
          "%4.1f FLOP/B" % (M, K, N, cnt, t1, f / t1 / 1e6, by / t1 / 1e6, t2, t3, hbm_us, mfma_us, f / by))
    tot[0] += cnt * t1
    tot[1] += cnt * t2
    tot[2] += cnt * t3
    tot[3] += cnt * max(hbm_us, mfma_us)
print("net totals (x occurrences): fwd %.1f us  dgrad %.1f us  wgrad %.1f us   sum of per-shape floors %.1f us" % tuple(tot))
