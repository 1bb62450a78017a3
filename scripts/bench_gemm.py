"""Per-layer GEMM micro-benchmark (B=1024 shapes of SURVEY 8a a10): TFLOP/s of the NN forward (with BN
stats), NN dgrad and TN wgrad launches, HIP events on the launch stream."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speech_recognition_amd import _lib
lib = _lib.load()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
shapes = [(397,128,128),(199,128,192),(197,192,192),(99,192,256),(97,256,256),(49,256,320),(47,320,320),(24,320,384),(22,384,384),(11,384,512),(9,512,512)]
S = _lib.stream_ptr()
def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
tot = [0, 0, 0]; fl = 0
for L, K, N in shapes:
    M = B * L
    A = torch.randn(M, K, device='cuda'); W = torch.randn(K, N, device='cuda') * 0.1; C = torch.empty(M, N, device='cuda')
    G = torch.randn(M, N, device='cuda'); WT = W.t().contiguous(); DZ = torch.empty(M, K, device='cuda'); dW = torch.empty(K, N, device='cuda')
    part = torch.empty(lib.kws_gemm_num_row_tiles(M) * 2 * N, device='cuda')
    ws = torch.empty(int(lib.kws_gemm_tn_workspace_floats(M, K, N)), device='cuda')
    t1 = timeit(lambda: _lib.call("kws_gemm_nn_f32", _lib.ptr(A), _lib.ptr(W), _lib.ptr(C), M, K, N, _lib.ptr(part), S))
    t2 = timeit(lambda: _lib.call("kws_gemm_nn_f32", _lib.ptr(G), _lib.ptr(WT), _lib.ptr(DZ), M, N, K, None, S))
    t3 = timeit(lambda: _lib.call("kws_gemm_tn_f32", _lib.ptr(A), _lib.ptr(G), _lib.ptr(dW), M, K, N, _lib.ptr(ws), S))
    f = 2.0 * M * K * N
    print("M=%7d K=%3d N=%3d  fwd %6.1f us %6.1f TF | dgrad %6.1f us %6.1f TF | wgrad %6.1f us %6.1f TF  (S*KN=%.1f MB)" % (
        M, K, N, t1 * 1e3, f / t1 / 1e9, t2 * 1e3, f / t2 / 1e9, t3 * 1e3, f / t3 / 1e9, ws.numel() * 4 / 1e6))
    tot[0] += t1; tot[1] += t2; tot[2] += t3; fl += f
print("total fwd %.3f ms (%.1f TF)  dgrad %.3f ms (%.1f TF)  wgrad %.3f ms (%.1f TF)" % (tot[0], fl / tot[0] / 1e9, tot[1], fl / tot[1] / 1e9, tot[2], fl / tot[2] / 1e9))
