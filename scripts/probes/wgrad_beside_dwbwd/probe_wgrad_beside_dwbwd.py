"""Round 5 probe: the weight-gradient GEMM of a layer (MFMA bound, off the backward chain) BESIDE the two
depthwise-backward passes of the same layer (HBM bound, on the chain), one grid, the CUs partitioned
(kws_dwconv_bwd_bn_wgrad_f32: the first G workgroups stream the pass, the others take weight-gradient work items).

Per layer of the raw-waveform net at batch 1024:
  seq        the step's schedule of round 4, as separate launches: all items, pass 1, fold, pass 2
  alone      each piece by itself
  tA(G, f)   ONE launch: pass 1 on G workgroups + the first 256 - G items, each run over the window [0, f) of its stages
             (f = 0: the pass alone on G CUs; f = 1: whole items) - where the row stops being flat the items take longer than the pass
  tB(G, f)   the same with pass 2 and the next 256 - G items
  rest       what is left of the items after the (G, f) the library's own plan picks (overlap_plan in csrc/net.hip restated here),
             resumed from the parked accumulators and run alone: seq - (tA + fold + tB + rest) is what the cut buys when the rest
             costs what it costs alone (in the step it rides in the next layer's input-gradient launch)
  co-run     two streams with NO events: 20 x {pass 1, fold, pass 2} on one, 20 x {all items} on the other, time to drain both / 20
Every fused / cut launch is first checked bit for bit against the separate launches (partial rows, dy, slabs).
(profiles/r05_wgrad_beside_dwbwd_whole_items.txt is the first form of this probe: whole items only, cut by item range.)"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speech_recognition_amd import _lib

lib = _lib.load()
B = int(os.environ.get("PROBE_B", "1024"))
REPS = int(os.environ.get("PROBE_REPS", "20"))
# (L_in, C = K, stride, N)
layers = [(399, 128, 1, 128), (397, 128, 2, 192), (199, 192, 1, 192), (197, 192, 2, 256), (99, 256, 1, 256), (97, 256, 2, 320),
          (49, 320, 1, 320), (47, 320, 2, 384), (24, 384, 1, 384), (22, 384, 2, 512), (11, 512, 1, 512)]
GS = [64, 96, 128, 160]
FS = [0.0, 0.125, 0.25, 0.375, 0.5, 0.625, 0.75, 0.875, 1.0]
main = torch.cuda.current_stream()
S = _lib.stream_ptr(main)
side_own = _lib.OwnedStream(torch.device("cuda"), 0)
side = side_own.stream
S2 = _lib.stream_ptr(side)


def timeit(fn, n=REPS):
    fn(); fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(main)
    for _ in range(n):
        fn()
    b.record(main); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def plan(B, Lin, Lout, K, N, items):
    """csrc/net.hip overlap_plan with its default constants"""
    import math
    M = B * Lout
    G = 128
    if items < 2 * (256 - G):
        G = 256 - (items // 2) // 8 * 8
    cus = 256 - G
    t_item = 2.0 * M * K * N / items / 0.41e6
    by1 = 4.0 * B * K * (Lin + Lout); by2 = 4.0 * B * K * (2 * Lin + Lout)
    t1 = by1 / (min(5800.0, G * 36.0) * 1e3); t2 = by2 / (min(5800.0, G * 50.0) * 1e3)

    def cut(t, avail):
        r = max(1, math.ceil(t / t_item)); cnt = min(avail, cus * r) // 8 * 8
        if cnt <= 0:
            return 0, 0
        f = int(0.9 * t * cus / (cnt * t_item) * 1024)
        if f >= 940:
            f = 1024
        if f < 32:
            return 0, 0
        return cnt, f
    nA, fA = cut(t1, items); nB, fB = cut(t2, items - nA)
    return G, nA, fA, nB, fB


tot = {"seq": 0.0, "cut": 0.0, "corun": 0.0, "tn": 0.0, "p1": 0.0, "p2": 0.0}
print("batch %d, %d repetitions per figure; us" % (B, REPS), flush=True)
for li, (Lin, C, s, N) in enumerate(layers):
    if s == 1:
        Lout, pad = Lin - 2, 0
    else:
        Lout = (Lin + 1) // 2
        pad = max((Lout - 1) * 2 + 3 - Lin, 0) // 2
    K = C
    M = B * Lout
    g = torch.Generator(device="cuda"); g.manual_seed(li)
    y = torch.randn(B, Lin, C, device="cuda", generator=g)
    w = torch.randn(3, C, device="cuda", generator=g)
    bn = torch.cat([torch.ones(C), torch.zeros(C), torch.zeros(C), torch.ones(C)]).cuda()
    dz = torch.randn(B, Lout, C, device="cuda", generator=g) * 1e-3
    z = torch.randn(M, K, device="cuda", generator=g)
    dY = torch.randn(M, N, device="cuda", generator=g) * 1e-3
    coef = torch.randn(2 * C, device="cuda", generator=g) * 1e-3
    dy = torch.empty(B, Lin, C, device="cuda"); dy2 = torch.empty_like(dy)
    nparts = int(lib.kws_dwconv_bwd_part_floats(B, Lin, C))
    part = torch.zeros(nparts, device="cuda"); part2 = torch.zeros(nparts, device="cuda")
    wsf = int(lib.kws_gemm_tn_workspace_floats(M, K, N))
    ws = torch.zeros(wsf, device="cuda"); ws2 = torch.zeros(wsf, device="cuda")
    ck = torch.zeros(int(lib.kws_gemm_tn_ckpt_floats(M, K, N)), device="cuda")
    dwg = torch.empty(3, C, device="cuda"); dgam = torch.empty(C, device="cuda"); dbet = torch.empty(C, device="cuda")
    coef_out = torch.empty(2 * C, device="cuda"); red = torch.empty(5 * C * 64, device="cuda")
    gran = ctypes.c_int(0)
    items = int(lib.kws_gemm_tn_items(M, K, N, ctypes.byref(gran)))
    gran = gran.value
    assert items > 0 and items % 8 == 0, (items, gran)
    Sout = ctypes.c_int(0)

    def fused(pas, G, lo, hi, f0=0, f1=1024, resume=(), st=S, part_=part, dy_=dy, ws_=ws):
        wi = _lib.WgradItems()
        wi.Z = z.data_ptr(); wi.dY = dY.data_ptr(); wi.M = M; wi.K = K; wi.N = N; wi.slabs = ws_.data_ptr(); wi.ckpt = ck.data_ptr()
        wi.item_lo = lo; wi.item_hi = hi; wi.f0 = f0; wi.f1 = f1; wi.n_resume = len(resume)
        for i, (rl, rh, rf) in enumerate(resume):
            wi.resume_lo[i] = rl; wi.resume_hi[i] = rh; wi.resume_f[i] = rf
        _lib.call("kws_dwconv_bwd_bn_wgrad_f32", _lib.ptr(dz), _lib.ptr(y), _lib.ptr(bn), _lib.ptr(w),
                  _lib.ptr(coef) if pas == 2 else None, _lib.ptr(dy_) if pas == 2 else None, _lib.ptr(part_) if pas == 1 else None,
                  pas, B, Lin, Lout, C, s, pad, ctypes.byref(wi), G, ctypes.byref(Sout), st)

    def p1(st=S, part_=part):
        _lib.call("kws_dwconv_bwd_bn_f32", _lib.ptr(dz), _lib.ptr(y), _lib.ptr(bn), _lib.ptr(w), None, None, _lib.ptr(part_), 1, B, Lin, Lout, C, s, pad, st)

    def p2(st=S, dy_=dy):
        _lib.call("kws_dwconv_bwd_bn_f32", _lib.ptr(dz), _lib.ptr(y), _lib.ptr(bn), _lib.ptr(w), _lib.ptr(coef), _lib.ptr(dy_), None, 2, B, Lin, Lout, C, s, pad, st)

    def fin(st=S):
        _lib.call("kws_dw_bwd_finalize", _lib.ptr(part), nparts // (5 * C), B * Lin, C, _lib.ptr(dwg), _lib.ptr(dgam), _lib.ptr(dbet),
                  _lib.ptr(coef_out), _lib.ptr(red), st)

    def tn_all(st=S, ws_=ws):
        fused(0, 0, 0, items, st=st, ws_=ws_)

    # ---- the library's own cut of this layer, checked bit for bit against the separate launches ----
    G0, nA, fA, nB, fB = plan(B, Lin, Lout, K, N, items)
    res = ((0, nA, fA if nA else 0), (nA, nA + nB, fB if nB else 0))

    def cut_A(**kw): fused(1, G0, 0, nA, 0, fA, **kw)
    def cut_B(**kw): fused(2, G0, nA, nA + nB, 0, fB, **kw)
    def cut_rest(**kw): fused(0, 0, 0, items, 0, 1024, resume=res, **kw)
    p1(part_=part); p2(dy_=dy); tn_all(ws_=ws)
    cut_A(part_=part2, ws_=ws2); cut_B(dy_=dy2, ws_=ws2); cut_rest(ws_=ws2)
    torch.cuda.synchronize()
    used = Sout.value * K * N
    assert torch.equal(part, part2), "pass 1 partial rows differ"
    assert torch.equal(dy, dy2), "pass 2 dy differs"
    assert torch.equal(ws[:used], ws2[:used]), "slabs differ"

    t_tn = timeit(tn_all); t_p1 = timeit(p1); t_fin = timeit(fin); t_p2 = timeit(p2)
    t_seq = timeit(lambda: (tn_all(), p1(), fin(), p2()))
    t_cutA = timeit(cut_A); t_cutB = timeit(cut_B); t_rest = timeit(cut_rest)
    t_cut = timeit(lambda: (cut_A(), fin(), cut_B(), cut_rest()))
    b1 = 4.0 * B * C * (Lin + Lout); b2 = 4.0 * B * C * (2 * Lin + Lout)
    fl = 2.0 * M * K * N
    print("\nL%d  L_in=%d C=K=%d s=%d N=%d  M=%d  items %d (granule %d, S=%d)" % (li, Lin, C, s, N, M, items, gran, Sout.value), flush=True)
    print("  alone: items %.1f (%.1f TFLOP/s, %.2f TB/s)  pass 1 %.1f (%.2f TB/s)  fold %.1f  pass 2 %.1f (%.2f TB/s)   seq %.1f" % (
        t_tn, fl / t_tn / 1e6, 4.0 * (M * K + M * N) / t_tn / 1e6, t_p1, b1 / t_p1 / 1e6, t_fin, t_p2, b2 / t_p2 / 1e6, t_seq), flush=True)
    print("  the plan: G=%d, pass 1 + %d items x %.3f: %.1f | pass 2 + %d items x %.3f: %.1f | the rest alone %.1f | in a row with the fold %.1f against seq %.1f: %+.1f" % (
        G0, nA, fA / 1024.0, t_cutA, nB, fB / 1024.0, t_cutB, t_rest, t_cut, t_seq, t_cut - t_seq), flush=True)
    for G in GS:
        n = 256 - G
        if 2 * n > items:
            continue
        rowA, rowB = [], []
        for f in FS:
            fi = int(f * 1024)
            rowA.append(timeit(lambda: fused(1, G, 0, n if fi else 0, 0, fi)))
            rowB.append(timeit(lambda: fused(2, G, n, 2 * n if fi else n, 0, fi)))
        print("  G=%3d  pass 1 + %3d items x f: %s   [alone on G: %.2f TB/s]" % (G, n, " ".join("%6.1f" % v for v in rowA), b1 / rowA[0] / 1e6), flush=True)
        print("         pass 2 + %3d items x f: %s   [alone on G: %.2f TB/s]" % (n, " ".join("%6.1f" % v for v in rowB), b2 / rowB[0] / 1e6), flush=True)
    # ---- two streams, no events: what any side-stream schedule is bounded by ----
    def corun():
        for _ in range(REPS):
            p1(); fin(); p2()
            tn_all(st=S2, ws_=ws2)
    corun(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0 = torch.cuda.Event(); e0.record(main); side.wait_event(e0)
    a.record(main)
    corun()
    e1 = torch.cuda.Event(); e1.record(side); main.wait_event(e1)
    b.record(main); torch.cuda.synchronize()
    t_co = a.elapsed_time(b) / REPS * 1e3
    print("  two streams, no events: %.1f us (%+.1f)" % (t_co, t_co - t_seq), flush=True)
    tot["seq"] += t_seq; tot["cut"] += t_cut; tot["corun"] += t_co; tot["tn"] += t_tn; tot["p1"] += t_p1; tot["p2"] += t_p2
    del y, dz, z, dY, dy, dy2, ws, ws2, ck
    torch.cuda.empty_cache()
print("\ntotals over the eleven layers (us): items alone %.1f, pass 1 %.1f, pass 2 %.1f; seq %.1f; the plan's cut (rest alone) %.1f (%+.1f); two streams no events %.1f (%+.1f)" % (
    tot["tn"], tot["p1"], tot["p2"], tot["seq"], tot["cut"], tot["cut"] - tot["seq"], tot["corun"], tot["corun"] - tot["seq"]), flush=True)
side_own.close()
