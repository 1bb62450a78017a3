"""Time-stretch kernel timing: KWS_LIB_PATH=variants/libkws_X.so python scripts/bench_stretch.py [B]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from speech_recognition_amd import tta  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
g = torch.Generator(device="cuda")
g.manual_seed(1)
x = (torch.randn((B, 16000), generator=g, device="cuda") * 0.0774).clamp_(-1, 1)
pcm = (x * 32767).to(torch.int16)


def timeit(fn, n=10):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


for name, fn in (("f32", lambda: tta.time_stretch(x, 0.9)), ("i16", lambda: tta.time_stretch(pcm, 0.9))):
    ms = timeit(fn)
    # 32 forward + 36 inverse real 2048-point FFTs per clip at 2.5 N log2 N flops + the vocoder
    gflop = B * 68 * 2.5 * 2048 * 11 / 1e9
    print("%s B=%d: %.3f ms  %.2f M clips/s  %.0f GB/s (in+out)  %.1f TFLOP/s (FFT only)" % (
        name, B, ms, B / ms / 1e3, B * (16000 * (4 if name == "f32" else 2) + 64000) / ms / 1e6, gflop / ms))
