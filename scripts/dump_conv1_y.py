"""Writes the first convolution's output (pre-BN y0, through kws_net_debug_view) and the BatchNorm table it feeds, for a few awkward
batch sizes, to an .npz - tests/test_conv1_variants_gpu.py compares two library builds bit for bit (KWS_LIB_PATH selects the build).
usage: dump_conv1_y.py <out.npz>"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from speech_recognition_amd import _lib  # noqa: E402
from speech_recognition_amd.net import DeviceNet  # noqa: E402

out = {}
for B in (1, 3, 70, 200, 1024):
    net = DeviceNet(_lib.KWS_NET_TS_ATTENTION, 12)
    net.initialize(seed=11)
    g = torch.Generator(device="cuda")
    g.manual_seed(B)
    x = (torch.randn((B, 16000), generator=g, device="cuda") * 0.0774).clamp_(-1, 1).contiguous()
    y = torch.eye(12, device="cuda")[torch.randint(0, 12, (B,), generator=g, device="cuda")].contiguous()
    probs = net.train_fwd_bwd(x, y, seed=5, step=1)
    torch.cuda.synchronize()
    out["y0_%d" % B] = net.debug_view(B, 0, 0).copy()
    out["bn0_%d" % B] = net.debug_view(B, 2, 0).copy()
    out["probs_%d" % B] = probs.cpu().numpy()
    p = net.predict(x)                                   # the inference form of the kernel (no statistics)
    out["pred_%d" % B] = p.cpu().numpy()
np.savez(sys.argv[1], **out)
print("lib:", _lib.LIB_PATH)
