"""Round 5 probe: a depthwise-backward pass and weight-gradient work items SHARING the CUs (two 384-thread workgroups per CU, one
of each kind: dwbwd_wgrad_share_kernel of a -DDW_BWD_THREADS=384 variant build) against PARTITIONING them (dwbwd_wgrad_kernel).
Layers whose weight-gradient plan uses 128 x 128 tiles.  t(f) = ONE launch: the pass + all 256 items over the stage window [0, f)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speech_recognition_amd import _lib
lib = _lib.load()
B, REPS = 1024, 20
layers = {0: (399, 128, 1, 128), 4: (99, 256, 1, 256), 8: (24, 384, 1, 384), 9: (22, 384, 2, 512), 10: (11, 512, 1, 512)}
FS = [0.0, 0.125, 0.25, 0.375, 0.5, 0.625, 0.75, 1.0]
main = torch.cuda.current_stream(); S = _lib.stream_ptr(main)
def timeit(fn, n=REPS):
    fn(); fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(main)
    for _ in range(n): fn()
    b.record(main); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for li, (Lin, C, s, N) in layers.items():
    Lout, pad = (Lin - 2, 0) if s == 1 else ((Lin + 1) // 2, max(((Lin + 1) // 2 - 1) * 2 + 3 - Lin, 0) // 2)
    K, M = C, B * Lout
    y = torch.randn(B, Lin, C, device="cuda"); w = torch.randn(3, C, device="cuda")
    bn = torch.cat([torch.ones(C), torch.zeros(C), torch.zeros(C), torch.ones(C)]).cuda()
    dz = torch.randn(B, Lout, C, device="cuda") * 1e-3; z = torch.randn(M, K, device="cuda"); dY = torch.randn(M, N, device="cuda") * 1e-3
    coef = torch.randn(2 * C, device="cuda") * 1e-3; dy = torch.empty(B, Lin, C, device="cuda"); dy2 = torch.empty_like(dy)
    nparts = int(lib.kws_dwconv_bwd_part_floats(B, Lin, C)); part = torch.zeros(nparts, device="cuda"); part2 = torch.zeros(nparts, device="cuda")
    wsf = int(lib.kws_gemm_tn_workspace_floats(M, K, N)); ws = torch.zeros(wsf, device="cuda"); ws2 = torch.zeros(wsf, device="cuda")
    ck = torch.zeros(int(lib.kws_gemm_tn_ckpt_floats(M, K, N)), device="cuda")
    items = int(lib.kws_gemm_tn_items(M, K, N, None)); Sout = ctypes.c_int(0)
    def fused(pas, G, lo, hi, f1=1024, part_=part, dy_=dy, ws_=ws):
        wi = _lib.WgradItems(); wi.Z = z.data_ptr(); wi.dY = dY.data_ptr(); wi.M = M; wi.K = K; wi.N = N; wi.slabs = ws_.data_ptr(); wi.ckpt = ck.data_ptr()
        wi.item_lo = lo; wi.item_hi = hi; wi.f0 = 0; wi.f1 = f1
        _lib.call("kws_dwconv_bwd_bn_wgrad_f32", _lib.ptr(dz), _lib.ptr(y), _lib.ptr(bn), _lib.ptr(w), _lib.ptr(coef) if pas == 2 else None,
                  _lib.ptr(dy_) if pas == 2 else None, _lib.ptr(part_) if pas == 1 else None, pas, B, Lin, Lout, C, s, pad, ctypes.byref(wi), G, ctypes.byref(Sout), S)
    def p1(): _lib.call("kws_dwconv_bwd_bn_f32", _lib.ptr(dz), _lib.ptr(y), _lib.ptr(bn), _lib.ptr(w), None, None, _lib.ptr(part), 1, B, Lin, Lout, C, s, pad, S)
    def p2(): _lib.call("kws_dwconv_bwd_bn_f32", _lib.ptr(dz), _lib.ptr(y), _lib.ptr(bn), _lib.ptr(w), _lib.ptr(coef), _lib.ptr(dy), None, 2, B, Lin, Lout, C, s, pad, S)
    # bit-identity of the sharing form
    p1(); p2(); fused(0, 0, 0, items, ws_=ws)
    fused(1, -256, 0, items // 2 // 8 * 8, part_=part2, ws_=ws2); fused(2, -256, items // 2 // 8 * 8, items, dy_=dy2, ws_=ws2)
    torch.cuda.synchronize()
    used = Sout.value * K * N
    assert torch.equal(part, part2) and torch.equal(dy, dy2) and torch.equal(ws[:used], ws2[:used])
    t_tn = timeit(lambda: fused(0, 0, 0, items)); t_p1 = timeit(p1); t_p2 = timeit(p2)
    print("\nL%d K=%d N=%d items %d | items alone %.1f  pass 1 %.1f  pass 2 %.1f (stand-alone kernels, 384 threads)" % (li, K, N, items, t_tn, t_p1, t_p2), flush=True)
    for name, G in (("share, 256 + 256 workgroups", -256), ("partition, G = 128", 128)):
        for pas in (1, 2):
            row = [timeit(lambda: fused(pas, G, 0, items if f else 0, int(f * 1024))) for f in FS]
            print("  %-28s pass %d + all items x f: %s" % (name, pas, " ".join("%6.1f" % v for v in row)), flush=True)
    print("  %-28s            items x f: %s" % ("the items alone", " ".join("%6.1f" % timeit(lambda: fused(0, 0, 0, items if f else 0, int(f * 1024))) for f in FS[1:])), flush=True)
    del y, dz, z, dY, dy, dy2, ws, ws2, ck; torch.cuda.empty_cache()
