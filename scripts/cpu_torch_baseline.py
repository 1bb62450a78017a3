"""A second CPU number beside bench.py's cpu_baseline (the NumPy oracle): the same raw-waveform net written with
torch.nn.functional on the HOST cores (oneDNN / MKL, all threads), forward + backward + RMSprop at batch 64 and 256 -
the closest stand-in available here for the reference's TF-CPU path.  Model only: no augmentation or
feature extraction.  Measurement script, not product code."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from oracle.net import TimeSlicedAttentionNet  # noqa: E402  (layer table and initial weights only)


def forward(net, P, x, y):
    B = x.shape[0]
    h = F.pad(x, (10, 10)).unfold(1, 40, 20).permute(0, 2, 1)
    h = F.conv1d(h, P['conv1d_1/kernel'].permute(2, 1, 0), stride=2)

    def bn_relu6(h, idx):
        h = F.batch_norm(h, None, None, P['batch_normalization_%d/gamma' % idx], P['batch_normalization_%d/beta' % idx],
                         training=True, eps=1e-3)
        return torch.clamp(h, 0, 6)
    h = bn_relu6(h, 1)
    for i, blk in enumerate(net.blocks):
        w = P['depthwise_conv2d_%d/depthwise_kernel' % (i + 1)].reshape(3, blk['cin'])
        h = F.conv1d(F.pad(h, blk['pad']), w.t().unsqueeze(1), stride=blk['stride'], groups=blk['cin'])
        h = F.conv1d(h, P['conv1d_%d/kernel' % (i + 2)].reshape(blk['cin'], blk['cout']).t().unsqueeze(2))
        h = bn_relu6(h, i + 2)
    a = h.permute(0, 2, 1)
    fd = F.dropout(a.reshape(B, -1), 0.4)
    att = torch.softmax(fd @ P['dense_1/kernel'] + P['dense_1/bias'], dim=1)
    feat = F.dropout(torch.cat([(a * att[:, :, None]).max(dim=1).values, a.mean(dim=1)], dim=1), 0.4)
    p = torch.softmax(feat @ P['dense_2/kernel'], dim=1)
    ysm = y * 0.9 + 0.1 / y.shape[1]
    return -(ysm * torch.log_softmax(torch.log(torch.clamp(p, 1e-7, 1 - 1e-7)), dim=1)).sum(dim=1).mean()


def main():
    if len(sys.argv) > 1:
        torch.set_num_threads(int(sys.argv[1]))
    net = TimeSlicedAttentionNet(dtype=np.float32)
    P = {k: torch.tensor(v, requires_grad=True) for k, v in net.params.items()}
    opt = torch.optim.RMSprop(P.values(), lr=1e-3, alpha=0.9, eps=1e-8)
    out = {"threads": torch.get_num_threads()}
    for B in (64, 256):
        x = torch.randn(B, 16000) * 0.0774
        y = torch.eye(12)[torch.randint(0, 12, (B,))]
        times = []
        for i in range(6):
            t0 = time.time()
            opt.zero_grad()
            forward(net, P, x, y).backward()
            opt.step()
            times.append(time.time() - t0)
        out["B%d" % B] = {"s_per_step": min(times[1:]), "clips_per_s": B / min(times[1:])}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
