import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import layers as OL
from speech_recognition_amd import _lib
_lib.load()
S = _lib.stream_ptr
def run(B, Lin, stride, pad, C, seed=0):
    rng = np.random.RandomState(seed)
    y = rng.randn(B, Lin, C).astype(np.float32) * 2.0
    w = rng.randn(3, C).astype(np.float32)
    gamma = (1 + 0.1 * rng.randn(C)).astype(np.float32); beta = (0.5 * rng.randn(C)).astype(np.float32)
    y64 = y.astype(np.float64)
    pre, (mean, var, rstd) = OL.bn_train_fwd(y64, gamma.astype(np.float64), beta.astype(np.float64))
    a = OL.relu6(pre); scale = gamma * rstd
    bn = np.concatenate([scale, beta - mean * scale, mean, rstd]).astype(np.float32)
    Lout = OL.valid_len(Lin + pad[0] + pad[1], 3, stride)
    dz = rng.randn(B, Lout, C).astype(np.float32)
    da_ref, dw_ref = OL.dwconv_bwd(dz.astype(np.float64), a, w.astype(np.float64), stride, pad)
    g_ref = da_ref * OL.relu6_mask(pre); xhat = (y64 - mean) * rstd
    t = [torch.from_numpy(v).cuda() for v in (dz, y, bn, w)]
    n_part = int(_lib.load().kws_dwconv_bwd_part_floats(B, Lin, C))
    part = torch.full((n_part,), float("nan"), device="cuda"); g = torch.full((B, Lin, C), float("nan"), device="cuda")
    _lib.call("kws_dwconv_bwd_f32", _lib.ptr(t[0]), _lib.ptr(t[1]), _lib.ptr(t[2]), _lib.ptr(t[3]), _lib.ptr(g), _lib.ptr(part), B, Lin, Lout, C, stride, pad[0], S())
    p = part.cpu().numpy().reshape(-1, 5, C).astype(np.float64).sum(0)
    refs = [g_ref.sum((0,1)), (g_ref*xhat).sum((0,1)), dw_ref[0], dw_ref[1], dw_ref[2]]
    errs = [np.abs(p[q]-refs[q]).max()/np.abs(refs[q]).max() for q in range(5)]
    print(B, Lin, stride, C, "g err %.1e" % (np.abs(g.cpu().numpy()-g_ref).max()), "sum errs", " ".join("%.1e" % e for e in errs), "nparts", n_part//(5*C))
for cfg in [(6,199,1,(0,0),192),(6,197,2,(1,1),192),(6,399,1,(0,0),128),(37,199,1,(0,0),192),(5,199,1,(0,0),192),(6,49,1,(0,0),320),(8,199,1,(0,0),192)]:
    run(*cfg)
