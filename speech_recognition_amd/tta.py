"""Batched test-time augmentation on the device (reference make_submission.py:120-146, BASELINE config C5).

The reference runs 3 (or 6) `model.predict` passes over NumPy copies of the batch (identity,
np.roll(X, -1500, axis=1), 1.2*X [, slow, clip(1.1*slow), 0.9*slow]) and averages the softmaxes.  Here
the augmented copies are produced by `kws_tta_transform`, each goes through the inference network
program, and `kws_tta_combine` forms the average and the argmax - the batch never leaves HBM.
TTA inference is embarrassingly parallel over clips: under torch.distributed every rank takes a
contiguous range of the test set and no collective is needed ("replicas only")."""
import ctypes

import torch

from . import _lib
from .device_array import as_device_f32

# kinds of kws_tta_transform
IDENTITY, ROLL_LEFT_1500, LOUD_1_2, SLOW_LOUD_CLIP_1_1, QUIET_0_9 = 0, 1, 2, 3, 4


def predict_tta(model, X, X_slow=None):
    """Returns (probs [B, C] f32 CUDA tensor, argmax [B] int32 CUDA tensor).

    X: raw clips [B, 16000] (DeviceArray / tensor / array).  X_slow: optional time-stretched clips
    (make_submission.py `use_speed_tta`): adds the three slow terms and divides the SUM OF SIX by 10, the
    reference's scaling (make_submission.py:137-140; it does not change the argmax)."""
    net = model.net
    xd = as_device_f32(X, net.device)
    B, L = xd.shape
    s = _lib.stream_ptr()
    terms = []
    scratch = torch.empty_like(xd)

    def run(src, kind):
        if kind == IDENTITY:
            inp = src
        else:
            _lib.call("kws_tta_transform", _lib.ptr(src), _lib.ptr(scratch), B, L, kind, s)
            inp = scratch
        terms.append(net.predict(inp))       # each predict allocates its own [B, C] output

    run(xd, IDENTITY)
    run(xd, LOUD_1_2)
    run(xd, ROLL_LEFT_1500)
    divisor = 3.0
    if X_slow is not None:
        xs = as_device_f32(X_slow, net.device)
        run(xs, IDENTITY)
        run(xs, SLOW_LOUD_CLIP_1_1)
        run(xs, QUIET_0_9)
        divisor = 10.0
    C = terms[0].shape[1]
    ptrs = (ctypes.c_void_p * len(terms))(*[t.data_ptr() for t in terms])
    probs = torch.empty((B, C), dtype=torch.float32, device=net.device)
    amax = torch.empty(B, dtype=torch.int32, device=net.device)
    _lib.call("kws_tta_combine", ptrs, len(terms), divisor, _lib.ptr(probs), _lib.ptr(amax), B, C, s)
    return probs, amax


def shard_range(n_items):
    """[start, stop) of the test set owned by this rank (no data-path collective)."""
    from . import parallel
    w, r = parallel.world_size(), parallel.rank()
    per = (n_items + w - 1) // w
    return min(r * per, n_items), min((r + 1) * per, n_items)
