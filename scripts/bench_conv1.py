"""First-convolution kernels alone (conv1.hip vs the generic gathered GEMMs): one training step of the raw-waveform net
under rocprofv3 is the measurement; this script just runs a few steps without the generator's concurrent kernels.
usage: rocprofv3 --kernel-trace --stats -d out -- python3 scripts/bench_conv1.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from speech_recognition_amd import _lib  # noqa: E402
from speech_recognition_amd.keras_api import Model, RMSprop  # noqa: E402
from speech_recognition_amd.net import DeviceNet  # noqa: E402

B = 1024
net = DeviceNet(_lib.KWS_NET_TS_ATTENTION, 12)
net.initialize(seed=3)
model = Model(net, RMSprop())
g = torch.Generator(device="cuda")
g.manual_seed(1)
# ROTATE=N: N different input batches in turn (the bench step reads a batch the generator produced ten queue slots
# earlier, never the tensor of the previous step)
n_x = int(os.environ.get("ROTATE", "1"))
xs = [(torch.randn((B, 16000), generator=g, device="cuda") * 0.0774).clamp_(-1, 1) for _ in range(n_x)]
y = torch.eye(12, device="cuda")[torch.randint(0, 12, (B,), device="cuda")].contiguous()
row = torch.zeros(4, device="cuda")
for i in range(24):
    model._train_step_async(xs[i % n_x], y, row)
torch.cuda.synchronize()
