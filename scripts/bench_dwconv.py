"""Per-layer micro-benchmark of the depthwise kernels (a9, a15) at the batch-1024 shapes of the raw-waveform net:
forward (BN + ReLU6 on load), backward pass 1 (partial sums only) and pass 2 (dy written), algorithmic GB/s each."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speech_recognition_amd import _lib
lib = _lib.load()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
# (L_in, C, stride): stride 1 = VALID (L_out = L_in - 2), stride 2 = SAME (L_out = ceil(L_in / 2), TF right-biased pad)
layers = [(399, 128, 1), (397, 128, 2), (199, 192, 1), (197, 192, 2), (99, 256, 1), (97, 256, 2), (49, 320, 1), (47, 320, 2),
          (24, 384, 1), (22, 384, 2), (11, 512, 1)]
S = _lib.stream_ptr()
def timeit(fn, n=30):
    fn(); fn(); fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
tot = [0.0, 0.0, 0.0]; byt = [0.0, 0.0, 0.0]
for Lin, C, s in layers:
    if s == 1: Lout, pad = Lin - 2, 0
    else:
        Lout = (Lin + 1) // 2
        total = max((Lout - 1) * 2 + 3 - Lin, 0); pad = total // 2
    y = torch.randn(B, Lin, C, device='cuda'); w = torch.randn(3, C, device='cuda')
    bn = torch.cat([torch.ones(C), torch.zeros(C), torch.zeros(C), torch.ones(C)]).cuda()
    z = torch.empty(B, Lout, C, device='cuda'); dz = torch.randn(B, Lout, C, device='cuda') * 1e-3
    dy = torch.empty(B, Lin, C, device='cuda'); coef = torch.zeros(2 * C, device='cuda')
    part = torch.empty(int(lib.kws_dwconv_bwd_part_floats(B, Lin, C)), device='cuda')
    t0 = timeit(lambda: _lib.call("kws_dwconv_fwd_f32", _lib.ptr(y), _lib.ptr(bn), _lib.ptr(w), _lib.ptr(z), B, Lin, Lout, C, s, pad, S))
    t1 = timeit(lambda: _lib.call("kws_dwconv_bwd_bn_f32", _lib.ptr(dz), _lib.ptr(y), _lib.ptr(bn), _lib.ptr(w), None, None, _lib.ptr(part), 1, B, Lin, Lout, C, s, pad, S))
    t2 = timeit(lambda: _lib.call("kws_dwconv_bwd_bn_f32", _lib.ptr(dz), _lib.ptr(y), _lib.ptr(bn), _lib.ptr(w), _lib.ptr(coef), _lib.ptr(dy), None, 2, B, Lin, Lout, C, s, pad, S))
    b0 = 4.0 * B * C * (Lin + Lout); b1 = b0; b2 = 4.0 * B * C * (2 * Lin + Lout)
    print("L_in=%3d C=%3d s=%d  fwd %6.1f us %5.2f TB/s | bwd pass 1 %6.1f us %5.2f TB/s | pass 2 %6.1f us %5.2f TB/s   (%.0f MB / %.0f MB)" % (
        Lin, C, s, t0, b0 / t0 / 1e6, t1, b1 / t1 / 1e6, t2, b2 / t2 / 1e6, b0 / 1e6, b2 / 1e6))
    for i, (t, b) in enumerate([(t0, b0), (t1, b1), (t2, b2)]): tot[i] += t; byt[i] += b
print("total fwd %.1f us (%.2f TB/s)  pass 1 %.1f us (%.2f TB/s)  pass 2 %.1f us (%.2f TB/s)" % (
    tot[0], byt[0] / tot[0] / 1e6, tot[1], byt[1] / tot[1] / 1e6, tot[2], byt[2] / tot[2] / 1e6))
