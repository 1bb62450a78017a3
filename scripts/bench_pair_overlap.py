"""Round 4 experiment: what would filling the input-gradient GEMM's tail with the weight-gradient GEMM of the same layer buy?
The two are independent (dz = dy W^T, dW = z^T dy).  Three schedules per layer, batch 1024, HIP events around 10 repetitions:
  seq   both on one stream (what the training step does)
  two   dgrad on a HIGH-priority stream, wgrad on a LOW-priority one, started together, joined by events
The difference bounds what ONE launch holding both grids could save (no second ramp-up / drain)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speech_recognition_amd import _lib
lib = _lib.load()
B = 1024
shapes = [(397,128,128),(199,128,192),(197,192,192),(99,192,256),(97,256,256),(49,256,320),(47,320,320),(24,320,384),(22,384,384),(11,384,512),(9,512,512)]
dev = torch.device("cuda")
hi = torch.cuda.Stream(priority=-1)
lo_own = _lib.OwnedStream(dev, -1)
lo = lo_own.stream
main = torch.cuda.current_stream()
tot = {"seq": 0.0, "two": 0.0, "dgrad": 0.0, "wgrad": 0.0}
for L, K, N in shapes:
    M = B * L
    A = torch.randn(M, K, device='cuda'); W = torch.randn(K, N, device='cuda') * 0.1
    G = torch.randn(M, N, device='cuda'); WT = W.t().contiguous(); DZ = torch.empty(M, K, device='cuda'); dW = torch.empty(K, N, device='cuda')
    ws = torch.empty(int(lib.kws_gemm_tn_workspace_floats(M, K, N)), device='cuda')
    def dgrad(s): _lib.call("kws_gemm_nn_f32", _lib.ptr(G), _lib.ptr(WT), _lib.ptr(DZ), M, N, K, None, _lib.stream_ptr(s))
    def wgrad(s): _lib.call("kws_gemm_tn_f32", _lib.ptr(A), _lib.ptr(G), _lib.ptr(dW), M, K, N, _lib.ptr(ws), _lib.stream_ptr(s))
    def seq():
        dgrad(main); wgrad(main)
    def two():
        e0 = torch.cuda.Event(); e0.record(main)
        hi.wait_event(e0); lo.wait_event(e0)
        dgrad(hi); wgrad(lo)
        e1 = torch.cuda.Event(); e1.record(hi); e2 = torch.cuda.Event(); e2.record(lo)
        main.wait_event(e1); main.wait_event(e2)
    def only_d(): dgrad(main)
    def only_w(): wgrad(main)
    res = {}
    for name, fn in (("seq", seq), ("two", two), ("dgrad", only_d), ("wgrad", only_w), ("seq2", seq), ("two2", two)):
        fn(); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(main)
        for _ in range(10): fn()
        b.record(main); torch.cuda.synchronize()
        res[name] = a.elapsed_time(b) * 100.0
    print("M=%7d K=%3d N=%3d | dgrad %6.1f + wgrad %6.1f = %6.1f us | one stream %6.1f / %6.1f us | two streams (hi / lo priority) %6.1f / %6.1f us" % (
        M, K, N, res["dgrad"], res["wgrad"], res["dgrad"] + res["wgrad"], res["seq"], res["seq2"], res["two"], res["two2"]), flush=True)
    for k in ("dgrad", "wgrad"): tot[k] += res[k]
    tot["seq"] += min(res["seq"], res["seq2"]); tot["two"] += min(res["two"], res["two2"])
print("totals: dgrad %.3f + wgrad %.3f ms alone; one stream %.3f ms; two streams %.3f ms" % (tot["dgrad"] / 1e3, tot["wgrad"] / 1e3, tot["seq"] / 1e3, tot["two"] / 1e3))
lo_own.close()
