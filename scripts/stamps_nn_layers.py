"""Per-layer in-kernel stamps of the wave-specialised NN GEMM for the 22 launches of a batch-1024 training step
(eleven forward with BN statistics, eleven input-gradient without), library built with -DKWS_GEMM_STAMP:
    scripts/build_variant.sh wsstamp -DKWS_GEMM_STAMP
    KWS_LIB_PATH=variants/libkws_wsstamp.so python scripts/stamps_nn_layers.py > profiles/r04_nn_per_layer.txt
Per K-slab iteration of a workgroup (shader cycles, median over the workgroups): what MFMA wave 0 spends issuing its
MFMAs + fragment reads, waiting at the barrier, and in the tile epilogue work it carries (staging writes, BN sums);
what loader wave 0 spends in its turns; the kernel-only time the same launch takes (HIP events, 10 launches)."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from speech_recognition_amd import _lib
lib = _lib.load()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
shapes = [(397,128,128),(199,128,192),(197,192,192),(99,192,256),(97,256,256),(49,256,320),(47,320,320),(24,320,384),(22,384,384),(11,384,512),(9,512,512)]
S = _lib.stream_ptr()
raw = ctypes.CDLL(_lib.LIB_PATH)
has_stamps = hasattr(raw, "kws_debug_read_stamps")


def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def stamps(kb):
    buf = np.zeros((8192, 8), dtype=np.uint64)
    raw.kws_debug_read_stamps(buf.ctypes.data_as(ctypes.c_void_p))
    t = buf[:256].astype(np.float64)
    ok = t[:, 3] > 0
    t = t[ok]
    it = t[:, 3]
    wb = buf[512:768].astype(np.float64)[ok]
    return dict(wgs=len(t), iters=it.mean(), mma=np.median(t[:, 0] / it), bar=np.median(t[:, 1] / it), stage=np.median(t[:, 2] / it),
                lwrite=np.median(t[:, 6] / it), lissue=np.median(t[:, 7] / it), total=np.median(t[:, 4] / it),
                ghz=np.median(t[:, 4] / t[:, 5]) * 0.1, wave_bar=[np.median(wb[:, w] / it) for w in range(8)],
                kernel_cycles=np.median(t[:, 4]), life_us=(t[:, 5].min() / 100.0, np.median(t[:, 5]) / 100.0, t[:, 5].max() / 100.0))


tot = {"fwd": 0.0, "dgrad": 0.0}
fl = 0.0
print("# %s" % os.environ.get("KWS_LIB_PATH", "default library"))
print("# layer  launch  M K N | tile | us  TFLOP/s | per K-slab iteration: MFMA-issue  barrier-wait  epilogue | loader write / issue | total per iteration | clock | barrier wait by wave (0-3 MFMA, 4-5 loaders, 6-7 storers)")
for li, (L, K, N) in enumerate(shapes):
    M = B * L
    A = torch.randn(M, K, device='cuda'); W = torch.randn(K, N, device='cuda') * 0.1; C = torch.empty(M, N, device='cuda')
    G = torch.randn(M, N, device='cuda'); WT = W.t().contiguous(); DZ = torch.empty(M, K, device='cuda')
    part = torch.empty(lib.kws_gemm_num_row_tiles(M) * 2 * max(N, K), device='cuda')
    for name, fn, kk, nn in (("fwd", lambda: _lib.call("kws_gemm_nn_f32", _lib.ptr(A), _lib.ptr(W), _lib.ptr(C), M, K, N, _lib.ptr(part), S), K, N),
                             ("dgrad", lambda: _lib.call("kws_gemm_nn_f32", _lib.ptr(G), _lib.ptr(WT), _lib.ptr(DZ), M, N, K, None, S), N, K)):
        us = timeit(fn)
        f = 2.0 * M * K * N
        tot[name] += us
        line = "L%-2d %-5s M=%7d K=%3d N=%3d | %7.1f us %6.1f TF" % (li, name, M, kk, nn, us, f / us / 1e6)
        if has_stamps:
            s = stamps(0)
            line += " | wgs %3d iters/wg %5.1f | mma %5.0f bar %4.0f epi %4.0f | ld write %4.0f issue %4.0f | iter %5.0f cyc | %.2f GHz | %s" % (
                s["wgs"], s["iters"], s["mma"], s["bar"], s["stage"], s["lwrite"], s["lissue"], s["total"], s["ghz"],
                " ".join("%.0f" % v for v in s["wave_bar"]))
            line += " | MFMA-wave lifetime of a workgroup min/median/max %.1f/%.1f/%.1f us" % s["life_us"]
        print(line, flush=True)
    fl += 2.0 * M * K * N
print("total fwd %.3f ms (%.1f TF)  dgrad %.3f ms (%.1f TF)" % (tot["fwd"] / 1e3, fl / tot["fwd"] / 1e6, tot["dgrad"] / 1e3, fl / tot["dgrad"] / 1e6))
