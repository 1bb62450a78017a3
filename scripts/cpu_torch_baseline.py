"""A second CPU number beside bench.py's cpu_baseline (the NumPy oracle): the same raw-waveform net written with
torch.nn.functional on the HOST cores (oneDNN / MKL, all threads), forward + backward + RMSprop at batch 64 and 256 -
the closest stand-in available here for the reference's TF-CPU path.  Model only: no augmentation or
feature extraction.  Measurement script, not product code."""
import json
import os
import sys
import time

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402



class Net(object):
    """Layer table of conv_1d_time_sliced_with_attention (reference model.py:775-838) and Glorot-uniform weights."""

    def __init__(self, num_classes=12):
        spec = [(1, 128), (2, 192), (1, 192), (2, 256), (1, 256), (2, 320), (1, 320), (2, 384), (1, 384), (2, 512), (1, 512)]
        rng = np.random.RandomState(0)

        def glorot(shape, fi, fo):
            lim = np.sqrt(6.0 / (fi + fo))
            return rng.uniform(-lim, lim, shape).astype(np.float32)
        self.params = {'conv1d_1/kernel': glorot((3, 40, 128), 120, 384)}
        self.blocks = []
        cin, L = 128, 399
        for i, (stride, cout) in enumerate(spec):
            if stride == 1:
                pad, Lout = (0, 0), L - 2
            else:
                Lout = -(-L // 2)
                p = max((Lout - 1) * 2 + 3 - L, 0)
                pad = (p // 2, p - p // 2)
            self.blocks.append(dict(cin=cin, cout=cout, stride=stride, pad=pad))
            self.params['depthwise_conv2d_%d/depthwise_kernel' % (i + 1)] = glorot((1, 3, cin, 1), 3 * cin, 3)
            self.params['conv1d_%d/kernel' % (i + 2)] = glorot((1, cin, cout), cin, cout)
            cin, L = cout, Lout
        for i in range(1, 13):
            c = 128 if i == 1 else spec[i - 2][1]
            self.params['batch_normalization_%d/gamma' % i] = np.ones(c, np.float32)
            self.params['batch_normalization_%d/beta' % i] = np.zeros(c, np.float32)
        self.T, self.C = L, cin
        self.params['dense_1/kernel'] = glorot((L * cin, L), L * cin, L)
        self.params['dense_1/bias'] = np.zeros(L, np.float32)
        self.params['dense_2/kernel'] = glorot((2 * cin, num_classes), 2 * cin, num_classes)


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
    net = Net()
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
