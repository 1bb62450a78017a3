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


def predict_tta(model, X, X_slow=None, use_speed_tta=False):
    """Returns (probs [B, C] f32 CUDA tensor, argmax [B] int32 CUDA tensor).

    X: raw clips [B, 16000] (DeviceArray / tensor / array).  X_slow: optional time-stretched clips
    (make_submission.py `use_speed_tta`): adds the three slow terms and divides the SUM OF SIX by 10, the
    reference's scaling (make_submission.py:137-140; it does not change the argmax).  With
    `use_speed_tta` and no X_slow the slow clips are made here by `time_stretch(X, 0.9)` instead of being
    read from the offline set of create_tta_set.py."""
    net = model.net
    xd = as_device_f32(X, net.device)
    if use_speed_tta and X_slow is None:
        X_slow = time_stretch(xd, 0.9, keep=xd.shape[1], device=net.device)
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


_PLANS = {}


def _stretch_plan(n_samples, rate):
    key = (int(n_samples), float(rate))
    if key not in _PLANS:
        plan = ctypes.c_void_p()
        _lib.check(_lib.load().kws_stretch_plan_create(key[0], key[1], ctypes.byref(plan)), "kws_stretch_plan_create")
        _PLANS[key] = plan
    return _PLANS[key]


def time_stretch(X, rate=0.9, keep=16000, wav_round_trip=True, device=None, in_scale=32768.0 / 32767.0):
    """The reference's offline slow set (create_tta_set.py:9-22), on the device and batched:
    librosa.effects.time_stretch(pcm / 32767, rate)[-keep:], and with `wav_round_trip` the int16 file the
    script writes as make_submission.py reads it back (np.int16(. * 32767) then DecodeWav's / 32768).

    X: [B, L] int16 PCM (torch tensor / array: the reference's `wavfile.read` input), or float clips in
    DecodeWav scale (int16 / 32768: DeviceArray / tensor / array), which are rescaled by `in_scale` = 32768 / 32767
    on load (pass 1.0 for clips that already are pcm / 32767).
    Returns a [B, keep] f32 CUDA tensor.  rate <= 0 raises (librosa: ParameterError)."""
    if rate <= 0:
        raise ValueError("rate must be a positive number")
    if isinstance(X, torch.Tensor) and X.dtype == torch.int16 or getattr(X, 'dtype', None) == 'int16':
        dev = device or (X.device if isinstance(X, torch.Tensor) and X.is_cuda else torch.device("cuda", torch.cuda.current_device()))
        xd = (X if isinstance(X, torch.Tensor) else torch.from_numpy(X)).to(dev).contiguous()
        i16 = True
    else:
        xd = as_device_f32(X, device or torch.device("cuda", torch.cuda.current_device()))
        i16 = False
    if xd.dim() == 1:
        xd = xd.reshape(1, -1)
    B, L = xd.shape
    plan = _stretch_plan(L, rate)
    out = torch.empty((B, keep), dtype=torch.float32, device=xd.device)
    with torch.cuda.device(xd.device):
        s = _lib.stream_ptr()
        if i16:
            _lib.call("kws_time_stretch_i16", plan, _lib.ptr(xd), _lib.ptr(out), B, keep, int(wav_round_trip), s)
        else:
            _lib.call("kws_time_stretch_f32", plan, _lib.ptr(xd), float(in_scale), _lib.ptr(out), B, keep,
                      int(wav_round_trip), s)
    return out


def stretched_samples(n_samples, rate):
    """Length of librosa 0.5's time_stretch output for an n_samples clip."""
    return _lib.load().kws_stretch_out_samples(_stretch_plan(n_samples, rate))


def shard_range(n_items):
    """[start, stop) of the test set owned by this rank (no data-path collective)."""
    from . import parallel
    w, r = parallel.world_size(), parallel.rank()
    per = (n_items + w - 1) // w
    return min(r * per, n_items), min((r + 1) * per, n_items)


def predict_test_set(model, clips, batch=4096, use_speed_tta=False, gather=True):
    """BASELINE config C5 as a driver: TTA inference over a whole test set (make_submission.py:86-146 walks its 158,538
    files in batches and averages the TTA terms per batch), sharded over the data-parallel ranks.

    clips: [n, 16000] host array / tensor holding the WHOLE set on every rank (or any indexable that yields a
    [k, 16000] block for a slice).  Rank r infers the contiguous range `shard_range(n)` in batches of `batch` clips -
    no collective on the data path.  With `gather` the per-rank results are exchanged once at the end (control plane:
    torch.distributed.all_gather_object) and every rank returns the full (probs [n, C], argmax [n]) as NumPy arrays;
    without it each rank returns (probs, argmax, (lo, hi)) for its own range."""
    import numpy as np
    from . import parallel
    n = len(clips)
    lo, hi = shard_range(n)
    probs, amax = [], []
    for s in range(lo, hi, batch):
        e = min(s + batch, hi)
        p, a = predict_tta(model, clips[s:e], use_speed_tta=use_speed_tta)
        probs.append(p.cpu().numpy())
        amax.append(a.cpu().numpy())
    C = model.net.num_classes
    p_loc = np.concatenate(probs) if probs else np.zeros((0, C), np.float32)
    a_loc = np.concatenate(amax) if amax else np.zeros((0,), np.int32)
    if not gather or not parallel.active():
        return (p_loc, a_loc) if gather else (p_loc, a_loc, (lo, hi))
    parts = [None] * parallel.world_size()
    parallel.dist.all_gather_object(parts, (lo, hi, p_loc, a_loc))
    parts.sort(key=lambda t: t[0])
    return np.concatenate([t[2] for t in parts]), np.concatenate([t[3] for t in parts])
