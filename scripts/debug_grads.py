import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np, torch
from test_net_gpu import _pair, _batch
B = int(sys.argv[1]) if len(sys.argv) > 1 else 6
ora, net = _pair()
x, y = _batch(B, 12, B)
probs = net.train_fwd_bwd(torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda(), seed=1234567, step=3)
torch.cuda.synchronize()
loss, p, grads, cache = ora.loss_and_grads(x.astype(np.float64), y.astype(np.float64), seed=1234567, step=3)
g = net.grads_dict()
for k, ref in grads.items():
    if k in ora.l2_names:
        ref = ref - 2e-5 * ora.params[k].astype(np.float64)
    ref = ref.reshape(g[k].shape)
    print("%-45s relerr %.2e  absmax %.3e" % (k, np.abs(g[k] - ref).max() / max(np.abs(ref).max(), 1e-12), np.abs(ref).max()))
# --- hypothesis: a ReLU6 kink flip (f32 vs f64 rounding of a pre-activation within ~1e-6 of 0 or 6) ---
for idx in range(1, 13):
    k = 'batch_normalization_%d/beta' % idx
    d = g[k] - grads[k]
    c = int(np.abs(d).argmax())
    yv, gam, stats, pre = cache['bn%d' % idx]
    prec = pre[..., c]
    near = np.minimum(np.abs(prec), np.abs(prec - 6.0))
    j = np.unravel_index(near.argmin(), near.shape)
    print("bn%d: max dbeta diff %.3e at c=%d; nearest-kink |pre|dist %.2e (pre=%.8f)  n(|dist|<1e-5)=%d" % (
        idx, d[c], c, near[j], prec[j], int((np.minimum(np.abs(pre), np.abs(pre - 6)) < 1e-5).sum())))
