"""Round 5 probe: can a depthwise-backward pass and the weight-gradient GEMM of the same layer SHARE the CUs (co-resident
workgroups: the matrix pipe for one, vector ALU + memory for the other) instead of partitioning them?  Two streams, NO events:
20 x {pass 1, fold, pass 2} on one stream, 20 x {weight-gradient GEMM (slabs only)} on the other, time to drain both / 20,
against the same launches in a row on one stream.  Run with the shipped library (512-thread depthwise workgroups: 2 x 168
registers per SIMD beside the GEMM's 2 x 136 do not fit -> the dispatcher partitions the CUs) and with a variant built with
-DDW_BWD_THREADS=256 -DDW_BWD2_THREADS=256 (one depthwise wave per SIMD: fits beside a 128 x 128 weight-gradient workgroup)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speech_recognition_amd import _lib
lib = _lib.load()
B, REPS = 1024, 20
layers = [(399, 128, 1, 128), (397, 128, 2, 192), (199, 192, 1, 192), (197, 192, 2, 256), (99, 256, 1, 256), (97, 256, 2, 320),
          (49, 320, 1, 320), (47, 320, 2, 384), (24, 384, 1, 384), (22, 384, 2, 512), (11, 512, 1, 512)]
main = torch.cuda.current_stream(); S = _lib.stream_ptr(main)
side_own = _lib.OwnedStream(torch.device("cuda"), 0); side = side_own.stream; S2 = _lib.stream_ptr(side)
def timeit(fn, n=REPS):
    fn(); fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(main)
    for _ in range(n): fn()
    b.record(main); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
tot = [0.0] * 4
for li, (Lin, C, s, N) in enumerate(layers):
    Lout, pad = (Lin - 2, 0) if s == 1 else ((Lin + 1) // 2, max(((Lin + 1) // 2 - 1) * 2 + 3 - Lin, 0) // 2)
    K, M = C, B * Lout
    y = torch.randn(B, Lin, C, device="cuda"); w = torch.randn(3, C, device="cuda")
    bn = torch.cat([torch.ones(C), torch.zeros(C), torch.zeros(C), torch.ones(C)]).cuda()
    dz = torch.randn(B, Lout, C, device="cuda") * 1e-3; z = torch.randn(M, K, device="cuda"); dY = torch.randn(M, N, device="cuda") * 1e-3
    coef = torch.randn(2 * C, device="cuda") * 1e-3; dy = torch.empty(B, Lin, C, device="cuda")
    nparts = int(lib.kws_dwconv_bwd_part_floats(B, Lin, C)); part = torch.zeros(nparts, device="cuda")
    ws = torch.zeros(int(lib.kws_gemm_tn_workspace_floats(M, K, N)), device="cuda")
    dwg = torch.empty(3, C, device="cuda"); dgam = torch.empty(C, device="cuda"); dbet = torch.empty(C, device="cuda")
    coef_out = torch.empty(2 * C, device="cuda"); red = torch.empty(5 * C * 64, device="cuda")
    items = int(lib.kws_gemm_tn_items(M, K, N, None)); dW = torch.empty(K, N, device="cuda")
    # the stand-alone weight-gradient kernel (6 - 8 waves, 80 - 136 registers) + its slab sum
    def tn(st=S): _lib.call("kws_gemm_tn_f32", _lib.ptr(z), _lib.ptr(dY), _lib.ptr(dW), M, K, N, _lib.ptr(ws), st)
    def p1(st=S): _lib.call("kws_dwconv_bwd_bn_f32", _lib.ptr(dz), _lib.ptr(y), _lib.ptr(bn), _lib.ptr(w), None, None, _lib.ptr(part), 1, B, Lin, Lout, C, s, pad, st)
    def p2(st=S): _lib.call("kws_dwconv_bwd_bn_f32", _lib.ptr(dz), _lib.ptr(y), _lib.ptr(bn), _lib.ptr(w), _lib.ptr(coef), _lib.ptr(dy), None, 2, B, Lin, Lout, C, s, pad, st)
    def fin(st=S): _lib.call("kws_dw_bwd_finalize", _lib.ptr(part), nparts // (5 * C), B * Lin, C, _lib.ptr(dwg), _lib.ptr(dgam), _lib.ptr(dbet), _lib.ptr(coef_out), _lib.ptr(red), st)
    t_tn = timeit(tn); t_p1 = timeit(p1); t_p2 = timeit(p2)
    t_seq = timeit(lambda: (tn(), p1(), fin(), p2()))
    def corun(dwf):
        for _ in range(REPS):
            dwf(); tn(S2)
    res = []
    for dwf in (lambda: (p1(), fin(), p2()), p1, p2):
        corun(dwf); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0 = torch.cuda.Event(); e0.record(main); side.wait_event(e0)
        a.record(main); corun(dwf)
        e1 = torch.cuda.Event(); e1.record(side); main.wait_event(e1)
        b.record(main); torch.cuda.synchronize()
        res.append(a.elapsed_time(b) / REPS * 1e3)
    print("L%-2d K=%3d N=%3d items %4d | GEMM %6.1f  pass 1 %5.1f  pass 2 %5.1f | in a row %6.1f | co-run: GEMM || {p1, fold, p2} %6.1f (%+6.1f)   GEMM || p1 %6.1f   GEMM || p2 %6.1f" % (
        li, K, N, items, t_tn, t_p1, t_p2, t_seq, res[0], res[0] - t_seq, res[1], res[2]), flush=True)
    tot[0] += t_seq; tot[1] += res[0]; tot[2] += t_tn; tot[3] += t_p1 + t_p2
    del y, dz, z, dY, dy, ws; torch.cuda.empty_cache()
print("totals: GEMM %.1f  passes %.1f  in a row %.1f  co-run %.1f (%+.1f)" % (tot[2], tot[3], tot[0], tot[1], tot[1] - tot[0]))
side_own.close()
