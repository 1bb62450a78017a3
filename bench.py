#!/usr/bin/env python
"""bench.py - clips/s of the keyword-spotting training hot path on N MI355X (one process per GPU).

Workload (BASELINE.json configs[1], SURVEY.md 8d "C2"): 12-class conv_1d_time_sliced_with_attention
net, batch 1024 synthetic 16000-sample fp32 clips per GPU.  One step = the reference's whole hot
path for one batch: sampler draw (reference RNG order) -> augment from the HBM clip bank ->
STFT+mel+DCT features (M=80, K=60) AND the raw waveform (the generator's 'mfcc_and_raw' output, i.e.
both arms of the A/B) -> forward + backward of the raw-waveform net -> [RCCL all-reduce] -> RMSprop.
Batches are produced by the AudioProcessor generator on a background thread with a depth-10 queue,
exactly like Keras fit_generator drives it.

Prints ONE JSON line (see DESIGN.md "Measurement" for every field).
"""
from __future__ import division, print_function

import argparse
import contextlib
import glob
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_HBM_GBS = 8000.0          # HBM3E spec (6.29 TB/s measured copy)
WANTED = 'stop down off right up go on yes left no'.split()
ALL30 = ('sheila nine stop bed four six down bird marvin cat off right seven eight up three happy go zero on '
         'wow dog yes five one tree house two left no').split()


def build_synthetic(device, n_bank, seed, L=16000, tone_amp=0.05, tone_step_hz=None, label_noise=0.0, n_val=4096):
    """SURVEY 8d synthetic inputs: x = 0.0774*N(0,1) clipped to [-1,1] + 0.05 sin(2 pi f_c t), f_c = 200(1+c) Hz;
    6 x 60 s noise recordings; label mix silence 13 % / unknown 60 % (train.py:40-45); a 'pseudo' partition.
    The val-acc parity run (scripts/val_acc_parity.py) asks for a task that does not saturate: a weaker tone (tone_amp), class
    frequencies tone_step_hz apart (f_c = 400 + step * c) and a share label_noise of ALL index entries (training, pseudo and
    validation) relabelled with a random word."""
    from speech_recognition_amd.input_data import ClipBank, SILENCE_LABEL
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    word_of_row = np.empty(n_bank, dtype=object)
    cls_of_row = np.zeros(n_bank, np.int64)
    n_wanted = n_bank // 2
    for r in range(n_bank):
        if r < n_wanted:
            w = WANTED[r % 10]
        else:
            w = [a for a in ALL30 if a not in WANTED][r % 20]
        word_of_row[r] = w
        cls_of_row[r] = 2 + WANTED.index(w) if w in WANTED else 1
    t = torch.arange(L, device=device, dtype=torch.float32) / 16000.0
    bank = torch.empty((n_bank, L), dtype=torch.float32, device=device)
    cls_t = torch.from_numpy(cls_of_row).to(device)
    for s in range(0, n_bank, 4096):
        e = min(s + 4096, n_bank)
        x = torch.randn((e - s, L), generator=g, device=device) * 0.0774
        f = 200.0 * (1.0 + cls_t[s:e].float()) if tone_step_hz is None else 400.0 + float(tone_step_hz) * cls_t[s:e].float()
        x += float(tone_amp) * torch.sin(2.0 * math.pi * f[:, None] * t[None, :])
        bank[s:e] = x.clamp_(-1.0, 1.0)
    rng = np.random.RandomState(59185)
    noise = [(rng.randn(960000) * 0.1).astype(np.float32) for _ in range(6)]
    cb = ClipBank(bank, noise, device)
    # Partitions are DISJOINT row ranges, as the reference's which_set hash split makes them (input_data.py:61-114,
    # train.py:40-45): validation never scores a clip the training or pseudo partition draws.
    n_pseudo = n_bank // 8
    n_val = min(n_val, n_bank // 8)
    n_val_w = n_val // 2                                             # half wanted words, half unknown words
    val_rows = list(range(n_wanted - n_val_w, n_wanted)) + list(range(n_bank - (n_val - n_val_w), n_bank))
    pseudo_rows = list(range(n_wanted - n_val_w - n_pseudo // 2, n_wanted - n_val_w))
    train_rows = list(range(n_wanted - n_val_w - n_pseudo // 2))
    unk_rows = list(range(n_wanted, n_bank - (n_val - n_val_w)))
    n_sil = int(math.ceil(len(train_rows) * 13.0 / 100))
    n_unk = min(int(math.ceil(len(train_rows) * 60.0 / 100)), len(unk_rows) - n_pseudo // 2)
    index = {
        'training': [(r, word_of_row[r]) for r in train_rows] + [(0, SILENCE_LABEL)] * n_sil +
                    [(r, word_of_row[r]) for r in unk_rows[:n_unk]],
        'pseudo': [(r, word_of_row[r]) for r in pseudo_rows] + [(r, word_of_row[r]) for r in unk_rows[n_unk:n_unk + n_pseudo // 2]],
        'validation': [(r, word_of_row[r]) for r in val_rows],
        'testing': [],
    }
    if label_noise > 0.0:
        lrng = np.random.RandomState(seed ^ 0x5EED)
        for part in ('training', 'pseudo', 'validation'):
            ent = index[part]
            for k in np.nonzero(lrng.rand(len(ent)) < label_noise)[0]:
                if ent[k][1] != SILENCE_LABEL:
                    ent[k] = (ent[k][0], ALL30[lrng.randint(len(ALL30))])
    return {'bank': cb, 'index': index}


def _blas_threads():
    try:
        from threadpoolctl import threadpool_info
        return int(max([p.get('num_threads', 1) for p in threadpool_info()] or [1]))
    except Exception:
        return os.cpu_count() or 1


def _cpu_clip(OF, rng, bank, noise):
    """One clip the way input_data.py:457-514 makes it: draws, x fg, roll, + noise x vol."""
    return OF.augment(bank[rng.randint(len(bank))], 1.0 + rng.uniform(-0.15, 0.15), rng.randint(-500, 1),
                      noise[rng.randint(0, 960000 - 16000):][:16000], rng.uniform(0, 0.15))


def feature_error_vs_oracle(proc, n_clips=16):
    """CHECKER leg (oracle/ is test infrastructure: it is only ever the yardstick here).  Max |device - float64 oracle| of the
    STFT+mel stage on clips of this run's own generator: |X|, log-mel (input_data.py:367-378) and the MFCC rows the
    generator hands out (input_data.py:379-381), plus the largest magnitudes, so the tolerances of tests/test_kernels_gpu.py
    can be read against what the shipped kernel (fp16-split first radix-16 pass and DCT, v_log_f32) really does."""
    from oracle import features as OF
    X, _ = proc.get_data(n_clips, 0, 0.3, 0.15, 0.3, 0.15, 0.3, [-500, 0], 'training', None, pseudo_frequency=0.6)
    raw = (X[1] if isinstance(X, (list, tuple)) else X).tensor
    proc._stream.synchronize()
    dev = {k: proc._features(raw, kind) for k, kind in (("mfcc", 0), ("spectrogram", 1), ("log_mel", 2))}
    proc._stream.synchronize()
    tables = OF.tables_path_b(480, proc._n_mel, proc._n_out)
    x64 = raw.cpu().numpy().astype(np.float64)
    mag, logmel, feat = OF.features(x64, tables, 160, dtype=np.float64, return_all=True)
    ref = {"mfcc": feat, "spectrogram": mag, "log_mel": logmel}
    out = {"clips": int(n_clips), "what": "max |device - float64 oracle| on %d clips of the run's own generator (M=%d, K=%d)"
                                          % (n_clips, proc._n_mel, proc._n_out)}
    for k in ("spectrogram", "log_mel", "mfcc"):
        d = dev[k].cpu().numpy().astype(np.float64).reshape(ref[k].shape)
        out[k] = {"max_abs_err_vs_f64": float(np.abs(d - ref[k]).max()), "max_abs_value": float(np.abs(ref[k]).max())}
    return out


def cpu_baselines(budget_s=30.0):
    """The CPU-baseline set of BASELINE.md section 2, timed on this box's host cores on bounded samples (the whole
    call is ~30 s).  All of it runs oracle/ code (kind "port": TensorFlow 1.4 cannot run here):

      B1  reference-style generator: ONE clip per call, float64 batch buffer, augment -> STFT -> |X| -> mel -> log ->
          DCT (input_data.py:457-536), single thread like the reference's generator;
      B2  the same features batched: scipy.fft.rfft over frame matrices, mel and DCT as one GEMM each, float32, cache-sized
          chunks on a pool of worker threads, at the best of a few worker counts (its GB/s stands next to the GPU STFT stage's);
      B3  the model step on CPU: forward + backward + optimizer of the 12-class raw-waveform net at batch 64 through
          torch-CPU (oneDNN; oracle/torch_net.py), Keras-SGD(momentum) and RMSprop, at the best of a few thread counts;
      B4  end to end: the B1 generator feeding B3 through a depth-10 queue (Keras fit_generator's default) - the number
          comparable with the GPU clips/s and with the reference's own logged ~200 clips/s.

    `value` is B4.  The all-NumPy oracle step that round 1 reported alone is kept as "numpy_port"."""
    import queue
    import threading
    from oracle import features as OF
    from oracle.net import TimeSlicedAttentionNet
    from oracle.torch_net import TorchTimeSlicedNet
    ncpu = os.cpu_count() or 1
    rng = np.random.RandomState(0)
    bank = (rng.randn(256, 16000) * 0.0774).astype(np.float32)
    noise = (rng.randn(960000) * 0.1).astype(np.float32)
    tables = OF.tables_path_b(480, 80, 60)
    share = budget_s / 6.0
    out = {}

    # ---- B1: per-clip generator, one thread --------------------------------------------------------------
    def gen_batch(B, with_features):
        data = np.zeros((B, 16000))
        feats = np.zeros((B, 98 * 60)) if with_features else None
        for i in range(B):
            clip = _cpu_clip(OF, rng, bank, noise)
            data[i, :] = clip
            if with_features:
                feats[i, :] = OF.features(clip, tables, 160, dtype=np.float32).reshape(-1)
        return data, feats
    for name, wf in (("raw", False), ("mfcc_80_60", True)):
        for _ in range(3):
            gen_batch(16, wf)      # warm-up: the first calls pay for the BLAS thread pool and page faults
        n, t0 = 0, time.time()
        while time.time() - t0 < share / 2:
            gen_batch(16, wf)
            n += 16
        out["B1_generator_" + name] = {"value": n / (time.time() - t0), "unit": "clips/s", "cores": 1,
                                       "sample": "%d clips, one per call" % n}
    # ---- B2: batched features: scipy.fft.rfft over [clips * 98, 512] frame matrices + one GEMM each
    # for mel and DCT, float32, in cache-sized chunks of 16 clips dealt to a pool of worker threads (BLAS pinned to one
    # thread per worker); the worker count is chosen like B3's thread count: the best of a few ---------------------------
    clips = np.stack([_cpu_clip(OF, rng, bank, noise) for _ in range(1024)])
    best2 = None
    try:
        from threadpoolctl import threadpool_limits
    except Exception:
        threadpool_limits = None
    for thr in sorted(set([min(ncpu, t) for t in (1, 8, 16, 32, 64)])):
        with (threadpool_limits(limits=1) if threadpool_limits else contextlib.nullcontext()):
            OF.features_batched(clips, tables, 160, workers=thr)
            n, t0 = 0, time.time()
            while n < 2 * len(clips) or time.time() - t0 < share / 10:
                OF.features_batched(clips, tables, 160, workers=thr)
                n += len(clips)
            rate = n / (time.time() - t0)
        if best2 is None or rate > best2[0]:
            best2 = (rate, thr, n)
    out["B2_batched_features"] = {"value": best2[0], "unit": "clips/s", "cores": best2[1],
                                  "GBps_algorithmic": best2[0] * (64000 + 98 * 60 * 4) / 1e9,
                                  "sample": "%d clips in batches of 1024: %d worker threads x chunks of 16 clips, scipy.fft.rfft over "
                                            "[16*98, 512] + mel GEMM + log + DCT GEMM, float32 (oracle.features.features_batched)"
                                            % (best2[2], best2[1])}
    # ---- B3: model step, torch-CPU ---------------------------------------------------------------------------
    B = 64
    x = (rng.randn(B, 16000) * 0.0774).astype(np.float32)
    y = np.eye(12, dtype=np.float32)[rng.randint(0, 12, B)]
    best = None
    for thr in sorted(set([min(ncpu, t) for t in (8, 16, 32)])):
        twin = TorchTimeSlicedNet(threads=thr)
        twin.init_optimizer('rmsprop')
        twin.train_step(x, y, 1e-3, dropout='torch')
        t0 = time.time()
        k = 0
        while k < 3 or (time.time() - t0 < share / 4 and k < 40):
            twin.train_step(x, y, 1e-3, step=k, dropout='torch')
            k += 1
        rate = B * k / (time.time() - t0)
        if best is None or rate > best[0]:
            best = (rate, thr, k)
    out["B3_model_step_rmsprop"] = {"value": best[0], "unit": "clips/s", "cores": best[1],
                                    "GFLOPs": best[0] * 0.337, "sample": "%d steps at batch 64, torch-CPU f32" % best[2]}
    thr = best[1]
    twin = TorchTimeSlicedNet(threads=thr)
    twin.init_optimizer('sgd')
    twin.train_step(x, y, 1e-2, dropout='torch')
    t0, k = time.time(), 0
    while k < 3 or (time.time() - t0 < share / 4 and k < 40):
        twin.train_step(x, y, 1e-2, step=k, dropout='torch')
        k += 1
    out["B3_model_step_sgd_momentum"] = {"value": B * k / (time.time() - t0), "unit": "clips/s", "cores": thr,
                                         "sample": "%d steps at batch 64, torch-CPU f32" % k}
    # ---- B4: generator thread -> depth-10 queue -> model step ------------------------------------------------
    twin = TorchTimeSlicedNet(threads=thr)
    twin.init_optimizer('rmsprop')
    q = queue.Queue(maxsize=10)
    stop = threading.Event()

    def producer():
        while not stop.is_set():
            d, _ = gen_batch(B, True)
            lab = np.eye(12, dtype=np.float32)[rng.randint(0, 12, B)]
            while not stop.is_set():
                try:
                    q.put((d.astype(np.float32), lab), timeout=0.1)
                    break
                except queue.Full:
                    pass
    th = threading.Thread(target=producer, daemon=True)
    th.start()
    d, lab = q.get()
    twin.train_step(d, lab, 1e-3, dropout='torch')
    t0, k = time.time(), 0
    while k < 3 or time.time() - t0 < 2 * share:
        d, lab = q.get()
        twin.train_step(d, lab, 1e-3, step=k, dropout='torch')
        k += 1
    dt = time.time() - t0
    stop.set()
    th.join(timeout=5)
    # ---- the all-NumPy oracle step (round 1's single number) -------------------------------------------------
    net = TimeSlicedAttentionNet(num_classes=12, dtype=np.float32)
    net.init_optimizer('rmsprop')
    t1, k2 = time.time(), 0
    while k2 < 1 or (time.time() - t1 < share and k2 < 4):
        net.train_step(x, y, 1e-3, seed=1, step=k2)
        k2 += 1
    out["numpy_port_model_step"] = {"value": B * k2 / (time.time() - t1), "unit": "clips/s", "cores": _blas_threads(),
                                    "sample": "%d steps at batch 64, NumPy f32 oracle" % k2}
    return {"value": B * k / dt, "unit": "clips/s", "cores": thr + 1, "kind": "port",
            "sample": "B4 end to end: %d train steps at batch 64 in %.1f s - single-thread per-clip generator (augment + "
                      "STFT/mel/DCT 80/60, oracle/features.py) feeding torch-CPU fwd/bwd/RMSprop (oracle/torch_net.py, %d "
                      "threads) through a depth-10 queue; host has %d logical cores" % (k, dt, thr, ncpu),
            "parts": out}


# (profiler family, device kernel, bound, what it is) - the dominant kernel first; each becomes one roofline object
ROOFLINE_KERNELS = [
    # round 4: a layer's input-gradient and weight-gradient GEMMs are ONE launch (gemm_dgrad_wgrad_kernel); the forward GEMMs
    # (and whatever the fused kernel does not take) stay gemm_nn_ws_kernel / gemm_tn_ws_kernel launches
    ("gemm_bwd_pair", "gemm_dgrad_wgrad_kernel", "mfma", "pointwise 1x1 convolutions: input gradient + weight gradient of a layer in one launch"),
    ("gemm_nn", "gemm_nn_ws_kernel", "mfma", "pointwise 1x1 convolutions: forward (+ input gradient where not paired)"),
    ("gemm_tn", "gemm_tn_ws_kernel", "mfma", "pointwise 1x1 convolutions: weight gradient (where not paired)"),
    ("conv1_fwd", "conv1_fwd_kernel", "mfma", "first convolution (frames of 40 hop 20, k3 s2) as a Toeplitz GEMM"),
    ("conv1_wgrad", "conv1_wgrad_slabsum_kernel", "mfma", "first convolution, weight gradient; since round 4 the same launch also sums the "
                                                          "slabs of the eleven pointwise weight gradients (137 MB, HBM bound) beside it"),
    ("stft_mel", "stft4_kernel", "hbm", "STFT 480/160/512 -> |X| -> mel 80 -> log -> DCT 60 (generator stream, low priority)"),
    ("augment", "augment_kernel", "hbm", "gather x foreground volume, circular roll, + noise x volume (generator stream)"),
    ("dwconv_fwd", "dwconv_fwd_kernel", "hbm", "depthwise k3 forward with BN+ReLU6 applied on load"),
    ("dwconv_bwd", "dwconv_bwd_kernel", "hbm", "depthwise k3 backward fused with the BatchNorm backward"),
]


# configs[2] (C3: conv_1d_log_mfcc at batch 2048): (profiler family, device kernels pooled in it, bound, what)
C3_ROOFLINE_KERNELS = [
    ("gemm_bwd_pair", "gemm_dgrad_wgrad_kernel", "mfma", "pointwise 1x1 convolutions: input gradient + weight gradient of a layer in one launch"),
    ("gemm_nn", ("gemm_nn_ws_kernel", "gemm_nn_persist_kernel"), "mfma", "pointwise forward, the three shortcut convolutions' input gradients, and the gathered "
                                                                          "first / shortcut convolutions (persistent kernel)"),
    ("gemm_tn", ("gemm_tn_kernel", "gemm_tn_ws_kernel"), "mfma", "weight gradients of the gathered convolutions (first, shortcuts)"),
    ("block_out_fwd", ("block_out_dw_fwd_kernel", "block_out_fwd_kernel"), "hbm",
     "residual join forward: max-pool(relu6(bn(y2))) + shortcut, and (round 6) the next block's first depthwise convolution in the same pass"),
    ("block_join_bwd", "block_join_bwd_kernel", "hbm", "residual join backward + BatchNorm backward, two passes (reductions, then dY)"),
    ("dwconv_fwd", "dwconv_fwd_kernel", "hbm", "depthwise k3 forward, BN + ReLU6 on load"),
    ("dwconv_bwd", "dwconv_bwd_kernel", "hbm", "depthwise k3 backward (+ BatchNorm backward of its input, two passes, where the input is a BN output)"),
    ("bn_finalize", ("bn_stats_finalize_kernel", "dw_bwd_finalize_kernel", "dw_grad_finalize_batch_kernel", "slice_reduce_kernel"), "hbm",
     "fixed-order folds of partial rows: 49 launches on the dependency chain (latency, not bandwidth)"),
    ("stft_mel", "stft4_kernel", "hbm", "STFT 480/160/512 -> |X| -> mel 40 -> log (40 x 98 per clip)"),
    ("slab_sum", "reduce_slabs_batch_kernel", "hbm", "batched slab sums of the weight gradients"),
    ("add", "add_strided_kernel", "hbm", "shortcut gradient added into the strided rows of the block input's gradient"),
]
# configs[4] (C5) plain inference
C5_ROOFLINE_KERNELS = [
    ("conv1_fwd", "conv1_fwd_kernel", "mfma", "first convolution"),
    ("dwconv_fwd", "dwconv_fwd_kernel", "hbm", "depthwise k3 forward, BN + ReLU6 on load"),
]


def load_pmc_traffic(pattern="r*_pmc_traffic.json"):
    """HBM bytes per launch from the committed rocprofv3 --pmc passes (scripts/pmc_traffic.py), newest round first.
    An entry is used only while the kernel's source file still hashes to what was profiled: a changed kernel reports
    traffic null instead of a stale number."""
    import glob
    import hashlib
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)), reverse=True):
        try:
            with open(path) as f:
                d = json.load(f)
        except Exception:
            continue
        cur = {}
        for src in d.get("sources", {}):
            try:
                with open(os.path.join(ROOT, "speech_recognition_amd", "csrc", src), "rb") as f:
                    cur[src] = hashlib.sha256(f.read()).hexdigest()[:16]
            except Exception:
                cur[src] = None
        d["_file"] = os.path.basename(path)
        d["_fresh"] = {src: cur.get(src) == h for src, h in d.get("sources", {}).items()}
        return d
    return None


def _source_hashes(names):
    import hashlib
    cur = {}
    for src in names:
        try:
            with open(os.path.join(ROOT, "speech_recognition_amd", "csrc", src), "rb") as f:
                cur[src] = hashlib.sha256(f.read()).hexdigest()[:16]
        except Exception:
            cur[src] = None
    return cur


def load_rocprof_stats(pattern="r*_kernel_stats_bench_b1024.csv"):
    """Average launch duration per device kernel from the newest committed `rocprofv3 --kernel-trace --stats` summary of the bench
    command (profiles/r0N_kernel_stats_bench_b1024.csv) - the SECOND clock next to this run's HIP events (the two
    differ by ~3 % across boxes; the line carries both).  Template instantiations of one kernel are pooled (total time / launches).
    Gated like the PMC traffic: the sidecar <csv>.sources.json (scripts/profile_round.sh) holds the hashes of the kernel sources at
    profiling time, and a kernel whose source file has changed since reports null."""
    import csv
    import re
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)), reverse=True):
        if "f16x2" in os.path.basename(path):
            continue
        try:
            agg = {}
            with open(path) as f:
                for r in csv.DictReader(f):
                    name = r["Name"].replace("void ", "").replace("(anonymous namespace)::", "")
                    m = re.match(r"(\w+)", name)
                    if not m:
                        continue
                    a = agg.setdefault(m.group(1), [0, 0.0])
                    a[0] += int(r["Calls"])
                    a[1] += float(r["TotalDurationNs"])
            side = path[:-4] + ".sources.json"
            sources = {}
            if os.path.exists(side):
                with open(side) as f:
                    sources = json.load(f)
            cur = _source_hashes(sources.get("sources", {}))
            return {"_file": os.path.basename(path), "kernels": {k: {"launches": c, "avg_us": t / c / 1e3} for k, (c, t) in agg.items() if c > 0},
                    "source_of": sources.get("source_of", {}),
                    "_fresh": {src: cur.get(src) == h for src, h in sources.get("sources", {}).items()}}
        except Exception:
            continue
    return None


def roofline_entry(prof, family, kernel, bound, what, pmc, rocprof=None):
    k = prof.get(family)
    if not k or k["ms"] <= 0 or k["count"] <= 0:
        return None
    sec = k["ms"] * 1e-3
    tflops = k["flops"] / sec / 1e12
    gbs = k["bytes"] / sec / 1e9
    traffic, src = None, None
    # a family may pool several device kernels (C3: the wave-specialised and the persistent NN kernel are both "gemm_nn"): traffic and
    # the rocprofv3 duration are then launch-weighted means over the kernels listed
    kernels = (kernel,) if isinstance(kernel, str) else tuple(kernel)
    kernel = kernels[0]
    if pmc is not None:
        recs = [pmc.get("kernels", {}).get(kn) for kn in kernels]
        recs = [r for r in recs if r is not None and r.get("launches_seen", 0) > 0]
        if recs:
            stale = sorted(set(r.get("source") for r in recs if not pmc["_fresh"].get(r.get("source"), False)))
            if not stale:
                n = sum(r["launches_seen"] for r in recs)
                traffic = sum(r.get("hbm_bytes_total", r["hbm_bytes_per_launch"] * r["launches_seen"]) for r in recs) / n
            src = "%s%s" % (pmc["_file"], "" if not stale else " (not used: %s changed since that PMC pass, or the pass predates source hashes)" % ", ".join(str(v) for v in stale))
    e = {"family": family, "kernel": kernel if len(kernels) == 1 else " + ".join(kernels), "what": what, "bound": bound,
         "launches": k["count"], "avg_launch_us": 1e3 * k["ms"] / k["count"],
         "algorithmic_bytes_per_launch": k["bytes"] / k["count"], "algorithmic_flops_per_launch": k["flops"] / k["count"],
         "traffic": traffic, "traffic_source": src,
         "frac_hbm": gbs / PEAK_HBM_GBS, "frac_flops": tflops / PEAK_F32_MFMA_TFLOPS}
    if bound == "mfma":
        e.update({"achieved": tflops, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": tflops / PEAK_F32_MFMA_TFLOPS})
    else:
        e.update({"achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": gbs / PEAK_HBM_GBS})
    # the same fraction by the OTHER clock: the committed rocprofv3 summary's average duration of this kernel (another box, another
    # run: `frac` above is this run's HIP events)
    e["frac_rocprof"], e["rocprof_avg_launch_us"], e["rocprof_source"] = None, None, None
    if rocprof is not None:
        recs = [(kn, rocprof["kernels"].get(kn)) for kn in kernels]
        recs = [(kn, r) for kn, r in recs if r is not None]
        if recs:
            stale = sorted(set(str(rocprof["source_of"].get(kn)) for kn, _ in recs if not rocprof["_fresh"].get(rocprof["source_of"].get(kn), False)))
            fresh = not stale
            e["rocprof_source"] = "%s%s" % (rocprof["_file"], "" if fresh else " (not used: %s changed since that profile, or it has no source hashes)" % ", ".join(stale))
            if fresh:
                us = sum(r["avg_us"] * r["launches"] for _, r in recs) / sum(r["launches"] for _, r in recs)
                per = (e["algorithmic_flops_per_launch"] / 1e12 / PEAK_F32_MFMA_TFLOPS) if bound == "mfma" else (e["algorithmic_bytes_per_launch"] / 1e9 / PEAK_HBM_GBS)
                e["rocprof_avg_launch_us"] = us
                e["frac_rocprof"] = per / (us * 1e-6)
    return e


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=15)
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--bank", type=int, default=65536)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--profile-steps", type=int, default=5)
    ap.add_argument("--no-ab", action="store_true",
                    help="skip the A/B legs: N=1 the same step with the pointwise GEMMs as fp16 x 2 split products (off by "
                         "default in the product); N>1 the same step with the gradient all-reduce split (--allreduce-split)")
    ap.add_argument("--allreduce-split", type=int, default=6,
                    help="N>1 A/B leg: block from which the gradients are all-reduced while the earlier blocks' backward "
                         "still runs (the headline step sends ONE buffer after the backward pass)")
    ap.add_argument("--n1-value", type=float, default=None,
                    help="clips/s of the N=1 run of the same build: fills scaling_vs_n1 = value / (N x n1-value)")
    ap.add_argument("--no-configs", action="store_true",
                    help="skip the `configs` legs (BASELINE configs[2] C3 and configs[4] C5 on this one GPU; N=1 only)")
    ap.add_argument("--wall-budget", type=float, default=360.0,
                    help="seconds after which the remaining CHECKER legs (A/B arms, configs, epoch anchor, val-acc parity) report "
                         "{'skipped': 'budget'} instead of starting; the timed region, its roofline and cpu_baseline always run")
    ap.add_argument("--no-val-acc", action="store_true",
                    help="skip the short val-acc parity run (scripts/val_acc_parity.py: the same batches trained on the "
                         "device and on the oracle's torch-CPU twin, val_acc next to val_acc_cpu); N=1 only")
    return ap.parse_args(argv)


def visible_gpu_count():
    """GPUs this process may use, counted WITHOUT the HIP runtime (the launcher below must not touch the GPU before it
    starts its ranks): KFD topology nodes with SIMDs (CPU nodes have simd_count 0), cut down by the *_VISIBLE_DEVICES
    lists.  Returns None when the topology cannot be read (then a CHILD process asks torch)."""
    n = 0
    paths = glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties")
    if not paths:
        return None
    for path in paths:
        try:
            with open(path) as f:
                props = dict(l.split()[:2] for l in f if len(l.split()) >= 2)
        except (IOError, OSError):
            continue
        if int(props.get("simd_count", "0")) > 0:
            n += 1
    # a container may see the whole host's topology but only its own GPUs' render nodes
    nodes = glob.glob("/dev/dri/renderD*")
    if nodes:
        n = min(n, len([d for d in nodes if os.access(d, os.R_OK | os.W_OK)]))
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([x for x in v.split(",") if x.strip() != ""]))
    return n


# ---- first contact of N > 1 ranks: RCCL has never joined more than one rank on this pool, so the very
# first collectives run under a watchdog BEFORE the 4.3 GB clip bank is built.  The watchdog is a fresh CHILD process that
# never touches the GPU (started before this process does): if the process group does not come up, or the all-reduce of
# ones + one gradient-sized (4.77 MB) all-reduce do not complete within the limit, the child prints the ONE JSON line of the
# run - with "error" and "stage" - on rank 0's stdout and kills rank 0 (the launcher then ends the other ranks and returns
# non-zero).  A hung collective may hold the GIL or sit inside the runtime: a thread of the hung process could not report it.
PREFLIGHT_WATCHDOG = r"""
import json, os, select, signal, sys, time
ppid, world, t_init, t_coll = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3]), float(sys.argv[4])
stage, limit = "init_process_group", t_init
deadline = time.time() + limit
buf = b""
while True:
    r, _, _ = select.select([0], [], [], max(0.0, deadline - time.time()))
    if not r:
        print(json.dumps({"metric": "1s 16kHz clips/sec training throughput", "value": None, "unit": "clips/s", "n_gpus": world,
                          "error": "rank 0 did not get through '%s' within %.0f s (hung collective / rendezvous): killed by the "
                                   "preflight watchdog" % (stage, limit), "stage": stage, "rccl_ranks": None}), flush=True)
        try:
            os.kill(ppid, signal.SIGKILL)
        except OSError:
            pass
        sys.exit(3)
    chunk = os.read(0, 256)
    if not chunk:
        sys.exit(0)          # rank 0 closed the pipe: it is past the preflight, or ended by itself and said why
    buf += chunk
    while b"\n" in buf:
        line, buf = buf.split(b"\n", 1)
        if line == b"start":
            stage, limit = "preflight all-reduce", t_coll
            deadline = time.time() + limit
        elif line == b"ok":
            sys.exit(0)
"""


class PreflightWatchdog(object):
    def __init__(self, json_out, world):
        import subprocess
        t_coll = float(os.environ.get("KWS_BENCH_PREFLIGHT_TIMEOUT", "60"))
        t_init = float(os.environ.get("KWS_BENCH_INIT_TIMEOUT", "600"))
        self.p = subprocess.Popen([sys.executable, "-c", PREFLIGHT_WATCHDOG, str(os.getpid()), str(world), str(t_init), str(t_coll)],
                                  stdin=subprocess.PIPE, stdout=json_out, close_fds=True)

    def say(self, word):
        try:
            self.p.stdin.write(word.encode() + b"\n")
            self.p.stdin.flush()
        except (IOError, OSError, ValueError):
            pass

    def done(self):
        self.say("ok")
        try:
            self.p.stdin.close()
            self.p.wait(timeout=10)
        except Exception:
            pass


def epoch_anchor(device, epochs=4, bank=65536):
    """BASELINE.md section 1's one comparable anchor, through the product path: the epoch of exp-195 (the model train.py builds at HEAD) -
    batch 384 (train.py:33), 95 training steps + 11 validation steps inside ConfusionMatrixCallback (4,224 clips; train.py:56-61,
    callbacks.py:45-83), ReduceLROnPlateau + TensorBoard + ModelCheckpoint(save_best_only) attached (train.py:62-68), driven by
    model.fit_generator (train.py:69-71).  logs_195 holds the wall time between consecutive epoch ends: 193.7 s median over 100 epochs on
    host "apple2" (GPU unknown), real speech_commands wavs decoded per clip by the reference's generator.  Here: the same loop on a
    synthetic, HBM-resident clip bank; seconds per epoch = the median difference between consecutive epoch ends (the first epoch, with its
    first-launch costs, reported separately)."""
    import shutil
    import tempfile
    from speech_recognition_amd.callbacks import ConfusionMatrixCallback
    from speech_recognition_amd.input_data import AudioProcessor, prepare_words_list
    from speech_recognition_amd.keras_api import Callback, ModelCheckpoint, ReduceLROnPlateau, TensorBoard
    from speech_recognition_amd.model import prepare_model_settings, speech_model
    from speech_recognition_amd.utils import data_gen
    B, steps, val_steps = 384, 95, 11
    spec = build_synthetic(device, bank, seed=195, n_val=val_steps * B)
    words = prepare_words_list(WANTED)
    settings = prepare_model_settings(label_count=len(words), sample_rate=16000, clip_duration_ms=1000, window_size_ms=30.0,
                                      window_stride_ms=10.0, dct_coefficient_count=80, num_log_mel_features=60,
                                      output_representation='raw')
    proc = AudioProcessor(spec, 13.0, 60.0, WANTED, 10.0, 0.0, settings, output_representation='raw', device=device)
    np.random.seed(195)
    train_gen = data_gen(proc, None, batch_size=B, mode='training', pseudo_frequency=0.6)
    val_gen = data_gen(proc, None, batch_size=B, mode='validation', pseudo_frequency=0.0)
    model = speech_model('conv_1d_time_sliced_with_attention', settings['desired_samples'], num_classes=settings['label_count'])

    class EpochEnds(Callback):            # LAST in the list: its on_epoch_end runs after the validation pass and the checkpoint write
        def __init__(self):
            Callback.__init__(self)
            self.t = []

        def on_train_begin(self, logs=None):
            torch.cuda.synchronize()
            self.t.append(time.time())

        def on_epoch_end(self, epoch, logs=None):
            torch.cuda.synchronize()
            self.t.append(time.time())
    ends = EpochEnds()
    tmp, cwd, old_stdout = tempfile.mkdtemp(prefix="kws_anchor_"), os.getcwd(), sys.stdout
    os.chdir(tmp)                         # ConfusionMatrixCallback writes its two text files into the cwd (callbacks.py:76-83)
    try:
        callbacks = [ConfusionMatrixCallback(val_gen, val_steps, wanted_words=words, all_words=words, label2int=proc.word_to_index),
                     ReduceLROnPlateau(monitor='val_categorical_accuracy', mode='max', factor=0.5, patience=4, verbose=0, min_lr=1e-5),
                     TensorBoard(log_dir='logs_210'),
                     ModelCheckpoint('checkpoints_210/ep-{epoch:03d}-vl-{val_loss:.4f}.hdf5', save_best_only=True,
                                     monitor='val_categorical_accuracy', mode='max'),
                     ends]
        hist = model.fit_generator(train_gen, steps_per_epoch=steps, epochs=epochs, verbose=0, callbacks=callbacks)
        files = sorted(os.listdir('checkpoints_210')) if os.path.isdir('checkpoints_210') else []
    finally:
        os.chdir(cwd)
        sys.stdout = old_stdout
        shutil.rmtree(tmp, ignore_errors=True)
        proc.close()
    d = [b - a for a, b in zip(ends.t[:-1], ends.t[1:])]
    steady = float(np.median(d[1:])) if len(d) > 1 else float(d[0])
    clips_epoch = (steps + val_steps) * B
    return {"what": "exp-195's epoch through the product path: batch 384, 95 training steps + 11 validation steps inside "
                    "ConfusionMatrixCallback, ReduceLROnPlateau + TensorBoard + ModelCheckpoint attached, model.fit_generator "
                    "(train.py:33,56-71); synthetic HBM-resident bank of %d clips" % bank,
            "epochs": epochs, "seconds_per_epoch": steady, "first_epoch_seconds": float(d[0]), "epoch_seconds": [float(v) for v in d],
            "clips_per_epoch": clips_epoch, "clips_per_s": clips_epoch / steady,
            "checkpoints_written": len(files), "val_categorical_accuracy": [float(v) for v in hist.history.get('val_categorical_accuracy', [])],
            "hardware": "1x MI355X, product path",
            "reference": {"seconds_per_epoch": 193.7, "source": "logs_195 wall time between epoch ends (median of 100 epochs), host apple2, GPU unknown; "
                                                                   "real speech_commands v0.01 wavs read and augmented per clip by the reference's "
                                                                   "Python generator (BASELINE.md section 1)",
                          "clips_per_s_derived": clips_epoch / 193.7},
            "epochs_per_s_ratio": 193.7 / steady}


def gather_per_rank(dist, value, world, device):
    """one float per rank -> the list of all ranks' values, in rank order, on every rank (dist None: a single rank)"""
    if not dist:
        return [float(value)]
    gathered = [torch.zeros(1, dtype=torch.float64, device=device) for _ in range(world)]
    dist.all_gather(gathered, torch.tensor([float(value)], dtype=torch.float64, device=device))
    return [float(g.item()) for g in gathered]


def preflight_collectives(dist, device, world, rank, n_floats=1191436):
    """The first two collectives of the run: an all-reduce of ones (how many ranks really joined) and ONE all-reduce of a
    buffer the size of the flat gradient buffer (1,191,436 floats = 4.77 MB), checked element-wise at both ends and by its sum."""
    t0 = time.time()
    if os.environ.get("KWS_BENCH_PREFLIGHT_FAIL") == "hang" and rank == 0:      # test hook: a collective that never returns
        time.sleep(3600)
    if os.environ.get("KWS_BENCH_PREFLIGHT_FAIL") == "raise":
        raise RuntimeError("KWS_BENCH_PREFLIGHT_FAIL=raise (test hook)")
    ones = torch.ones(1, dtype=torch.float32, device=device)
    dist.all_reduce(ones)
    joined = int(round(float(ones.item())))
    if joined != world:
        raise RuntimeError("all-reduce of ones returned %d on a world of %d" % (joined, world))
    buf = torch.full((n_floats,), float(rank + 1), dtype=torch.float32, device=device)
    dist.all_reduce(buf)
    torch.cuda.synchronize(device)
    expect = world * (world + 1) / 2.0
    got = (float(buf[0].item()), float(buf[-1].item()), float(buf.double().sum().item()) / n_floats)
    if any(abs(g - expect) > 1e-6 for g in got):
        raise RuntimeError("gradient-sized all-reduce returned %r, expected %r everywhere" % (got, expect))
    return {"rccl_ranks": joined, "grad_allreduce_floats": n_floats, "grad_allreduce_ok": True,
            "seconds": time.time() - t0, "watchdog_s": float(os.environ.get("KWS_BENCH_PREFLIGHT_TIMEOUT", "60"))}


def launch_ranks(args, json_out):
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves.

    This process never touches the GPU: devices are counted from the KFD topology in sysfs (visible_gpu_count; when that
    is unreadable a throw-away CHILD process asks torch), the ranks are CHILD processes under torch.distributed.run, their
    stderr is passed through, rank 0's single JSON line is relayed, the child's return code is ours.  Fewer than N
    visible devices is an error, never a silent 1-GPU run (KWS_BENCH_ONE_DEVICE, the 1-GPU test hook, waives the count)."""
    import socket
    import subprocess
    n_dev = visible_gpu_count()
    if n_dev is None:
        n_dev = int(subprocess.check_output([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"]).split()[-1])
    if n_dev < args.gpus and not os.environ.get("KWS_BENCH_ONE_DEVICE"):
        raise SystemExit("bench.py --gpus %d: only %d GPU(s) visible" % (args.gpus, n_dev))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"),
               KWS_BENCH_CHILD="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    child = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, cwd=ROOT)
    out_b, _ = child.communicate()
    lines = [l for l in out_b.decode().splitlines() if l.lstrip().startswith("{")]
    if child.returncode != 0 or len(lines) != 1:
        sys.stderr.write(out_b.decode())
        raise SystemExit(child.returncode or 1)
    base = None
    if not args.no_cpu_baseline:
        # the host-core baseline of the same run is measured HERE, after the ranks have finished: its BLAS threads
        # would otherwise compete with the ranks' launch threads inside the timed region
        try:
            base = cpu_baselines()
        except Exception as e:    # a broken checker must not lose the GPU measurement
            base = {"error": repr(e)}
    out = json.loads(lines[0])
    if not args.no_cpu_baseline:
        out["cpu_baseline"] = base
    print(json.dumps(out), file=json_out, flush=True)


def main():
    # stdout carries exactly ONE line (the JSON result); everything else the pipeline prints on the way
    # (data_gen's per-epoch "[Ep:...]" line mirrors the reference and goes to stdout) is sent to stderr
    json_out = sys.stdout
    sys.stdout = sys.stderr
    if os.environ.get("KWS_BENCH_TRACE"):        # debugging aid: dump every thread's stack after N seconds and exit
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["KWS_BENCH_TRACE"]), exit=True)
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args, json_out)
    t_start = time.time()
    skipped = []

    def in_budget(leg):
        """a checker leg starts only inside the wall budget (a slow host must not turn the checkers into a lost measurement at the
        driver's limit); what was skipped is listed in the line"""
        if time.time() - t_start < args.wall_budget:
            return True
        skipped.append(leg)
        sys.stderr.write("leg %s skipped: %.0f s wall budget spent\n" % (leg, args.wall_budget))
        return False

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("KWS_BENCH_ONE_DEVICE"):   # test hook: N ranks on ONE GPU over gloo (1-GPU boxes cannot run RCCL x N)
        local_rank = 0
    dog = PreflightWatchdog(json_out, world) if (world > 1 and rank == 0) else None    # before anything touches the GPU
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: there is no CPU fallback of the product path")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        try:
            if os.environ.get("KWS_BENCH_PREFLIGHT_FAIL") == "init":          # test hook: a rendezvous / communicator that raises
                raise RuntimeError("KWS_BENCH_PREFLIGHT_FAIL=init (test hook)")
            if os.environ.get("KWS_BENCH_ONE_DEVICE"):
                dist.init_process_group("gloo", rank=rank, world_size=world)
            else:
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        except Exception as ex:          # first contact can fail right here (no xGMI, IPC mode, ports): one JSON line, non-zero exit
            if rank == 0:
                print(json.dumps({"metric": "1s 16kHz clips/sec training throughput", "value": None, "unit": "clips/s", "n_gpus": world,
                                  "error": repr(ex), "stage": "init_process_group", "rccl_ranks": None}), file=json_out, flush=True)
            sys.stderr.write("init_process_group failed on rank %d: %r\n" % (rank, ex))
            if dog:
                dog.done()
            raise SystemExit(3)
    if world != args.gpus:
        raise SystemExit("bench.py --gpus %d started with WORLD_SIZE=%d" % (args.gpus, world))
    rccl_ranks = 1
    preflight = None
    if dist:      # first contact, under the watchdog, before the clip bank is built
        if dog:
            dog.say("start")
        try:
            preflight = preflight_collectives(dist, device, world, rank)
        except Exception as ex:
            if rank == 0:
                print(json.dumps({"metric": "1s 16kHz clips/sec training throughput", "value": None, "unit": "clips/s", "n_gpus": world,
                                  "error": repr(ex), "stage": "preflight all-reduce", "rccl_ranks": None}), file=json_out, flush=True)
            sys.stderr.write("preflight collectives failed on rank %d: %r\n" % (rank, ex))
            if dog:
                dog.done()
            raise SystemExit(3)
        rccl_ranks = preflight["rccl_ranks"]
        if dog:
            dog.done()

    from speech_recognition_amd import _lib
    from speech_recognition_amd.input_data import AudioProcessor
    from speech_recognition_amd.keras_api import GeneratorEnqueuer
    from speech_recognition_amd.model import prepare_model_settings, speech_model
    from speech_recognition_amd.utils import data_gen
    from speech_recognition_amd.input_data import prepare_words_list

    B = args.batch
    settings = prepare_model_settings(label_count=len(prepare_words_list(WANTED)), sample_rate=16000,
                                      clip_duration_ms=1000, window_size_ms=30.0, window_stride_ms=10.0,
                                      dct_coefficient_count=80, num_log_mel_features=60,
                                      output_representation='mfcc_and_raw')
    spec = build_synthetic(device, args.bank, seed=59185)
    proc = AudioProcessor(spec, 13.0, 60.0, WANTED, 10.0, 0.0, settings, output_representation='mfcc_and_raw',
                          device=device)
    np.random.seed(1234 + rank)
    gen = data_gen(proc, None, batch_size=B, mode='training', pseudo_frequency=0.6)
    model = speech_model('conv_1d_time_sliced_with_attention', settings['desired_samples'],
                         num_classes=settings['label_count'])
    model.seed = 87654321            # one dropout stream for the global batch: rank r uses rows [r*B, (r+1)*B)
    model.allreduce_split = 0        # the headline step: ONE all-reduce of the flat gradient buffer after the backward pass
    ab_steps = min(args.steps, 50)
    ring = torch.zeros((args.warmup + args.steps + 3 * args.profile_steps + 2 * (ab_steps + 8) + 4 * (ab_steps + 14) + 6 * (ab_steps + 4) + 8, 4), dtype=torch.float32, device=device)
    enq = GeneratorEnqueuer(gen, max_queue_size=10, device=device)
    enq.start()

    def step(i):
        (mfcc, raw), y = enq.get()
        mfcc.wait()                      # the STFT+mel arm was produced for this batch too
        model._train_step_async(raw, y, ring[i])

    def barrier():
        if dist:
            dist.barrier()
        torch.cuda.synchronize()

    def timed_steps(first, n):
        """n steps between barrier + synchronize on both sides -> (max over ranks, this rank's own seconds)"""
        barrier()
        t0 = time.time()
        for i in range(n):
            step(first + i)
        barrier()
        own = time.time() - t0
        if dist:
            tmax = torch.tensor([own], dtype=torch.float64, device=device)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            return float(tmax.item()), own
        return own, own

    for i in range(args.warmup):
        step(i)
    dt, dt_own = timed_steps(args.warmup, args.steps)
    clips = B * world * args.steps
    ms = ring[args.warmup:args.warmup + args.steps].cpu().numpy()
    # every rank's own wall time per step (a straggler shows here) and the seed its sampler drew from (ranks must not train on
    # the same clips)
    per_rank_ms = gather_per_rank(dist, 1e3 * dt_own / args.steps, world, device)
    sampler_seeds = [int(v) for v in gather_per_rank(dist, float(1234 + rank), world, device)]

    # ---- per-kernel durations of the same step, HIP events on the launch stream -------------------
    prof = None
    roof = None
    stages = []
    used = args.warmup + args.steps
    if args.profile_steps > 0:
        # EVERY rank runs the profiled steps (each step contains the gradient all-reduce: a rank that skipped them
        # would leave the others waiting in the collective); only rank 0 records and reports the kernel times.
        # The profiler is a handle: this thread attaches to it, the generator thread through proc.profiler.
        profiler = _lib.Profiler() if rank == 0 else None
        if profiler is not None:
            profiler.attach()
            proc.profiler = profiler
        for i in range(args.profile_steps):
            step(used + i)
        barrier()
        used += args.profile_steps
        if profiler is not None:
            proc.profiler = None
            profiler.detach()
            prof = profiler.collect()
            pmc = load_pmc_traffic()
            rocprof = load_rocprof_stats()
            for family, kernel, bound, what in ROOFLINE_KERNELS:
                e = roofline_entry(prof, family, kernel, bound, what, pmc, rocprof)
                if e is not None:
                    stages.append(e)
            # the line's `roofline` = the dominant kernel of the step: the GEMM family with the largest summed HIP-event time
            gemm_stages = [e for e in stages if e["family"] in ("gemm_bwd_pair", "gemm_nn")]
            roof = max(gemm_stages, key=lambda e: e["launches"] * e["avg_launch_us"]) if gemm_stages else None
            # the weight-gradient GEMMs' slab sums run as ONE batched launch per step ("slab_sum"): priced into their family
            ss = prof.get("slab_sum")
            for e in stages:
                if e["family"] in ("gemm_tn", "gemm_bwd_pair") and ss and ss["count"] > 0:
                    us = e["avg_launch_us"] + 1e3 * ss["ms"] / e["launches"]
                    e["slab_sum_us_per_step"] = 1e3 * ss["ms"] / ss["count"]
                    e["avg_launch_us_incl_slab_sum"] = us
                    e["frac_incl_slab_sum"] = e["algorithmic_flops_per_launch"] / (us * 1e-6) / 1e12 / PEAK_F32_MFMA_TFLOPS
    # ---- N > 1: what the gradient exchange costs, and the split-overlap form of the same step -----
    exchange = None
    ab = {}
    if dist:
        n_grad = int(model.net.n_params)
        buf = torch.zeros(n_grad, dtype=torch.float32, device=device)
        for _ in range(5):
            dist.all_reduce(buf)
        barrier()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n_ar = 50
        t0 = time.time()
        e0.record()
        for _ in range(n_ar):
            dist.all_reduce(buf)         # the collective of parallel.allreduce_grads: the current stream waits for each
        e1.record()
        e1.synchronize()
        host_us = 1e6 * (time.time() - t0) / n_ar
        ar_us = torch.tensor([1e3 * e0.elapsed_time(e1) / n_ar], dtype=torch.float64, device=device)
        dist.all_reduce(ar_us, op=dist.ReduceOp.MAX)
        exchange = {"what": "%d back-to-back all-reduces (sum, in place) of the flat gradient buffer alone, HIP events on "
                            "the launch stream, max over ranks" % n_ar,
                    "floats": n_grad, "bytes": 4 * n_grad, "us_per_allreduce": float(ar_us.item()),
                    "host_us_per_allreduce_rank0": host_us,
                    "pct_of_ms_per_step": 100.0 * float(ar_us.item()) * 1e-3 / (1e3 * dt / args.steps),
                    "algorithmic_bus_GBps": 2.0 * (world - 1) / world * 4 * n_grad / (float(ar_us.item()) * 1e-6) / 1e9}
        if not args.no_ab and 1 <= args.allreduce_split < 11:
            model.allreduce_split = args.allreduce_split
            for i in range(8):
                step(used + i)
            dt_ab, _ = timed_steps(used + 8, ab_steps)
            used += 8 + ab_steps
            model.allreduce_split = 0
            ab["ab_allreduce_split"] = {
                "what": "the same step with the gradients of blocks >= %d (+ tail) all-reduced while the earlier blocks' "
                        "backward still runs, the rest afterwards (bit-identical sums; Model.allreduce_split)" % args.allreduce_split,
                "split_block": args.allreduce_split, "steps": ab_steps, "ms_per_step": 1e3 * dt_ab / ab_steps,
                "value": B * world * ab_steps / dt_ab, "unit": "clips/s",
                "vs_one_buffer": (B * world * ab_steps / dt_ab) / (clips / dt)}
    # ---- N = 1 A/B leg: the same step with the pointwise GEMMs (forward, input gradient, weight
    # gradient) as three f16 MFMA products of scaled two-way operand splits; the line's value / roofline above are the
    # f32-MFMA path.  The arm is a property of the net handle (kws_net_set_gemm_mode).
    gemm_mode = model.net.gemm_mode
    if world == 1 and not args.no_ab and gemm_mode == 0:
        try:
            model.net.set_gemm_mode(2)
            for i in range(8):
                step(used + i)
            dt_ab, _ = timed_steps(used + 8, ab_steps)
            used += 8 + ab_steps
            # the arm's own GEMM kernels against BOTH roofs (HIP events on the launch stream, as for the product arm):
            # algorithmic bytes over 8 TB/s, algorithmic FLOPs over the f32-MFMA peak the product arm is priced against
            arm_stages = []
            if args.profile_steps > 0:
                profiler = _lib.Profiler()
                profiler.attach()
                for i in range(args.profile_steps):
                    step(used + i)
                barrier()
                used += args.profile_steps
                profiler.detach()
                prof_ab = profiler.collect()
                for fam, kern, what_k in (("gemm_nn_f16x2", "gemm_nn_f16x2_kernel", "pointwise 1x1 convolutions: forward + input gradient"),
                                          ("gemm_tn_f16x2", "gemm_tn_f16x2_kernel", "pointwise 1x1 convolutions: weight gradient")):
                    e = roofline_entry(prof_ab, fam, kern, "hbm", what_k, load_pmc_traffic("r*_f16x2_arm_pmc.json"))
                    if e is not None:
                        arm_stages.append(e)
            ab["ab_gemm_f16x2"] = {"what": "pointwise forward / input-gradient / weight-gradient GEMMs as power-of-two scaled 2-way "
                                           "fp16 splits (three f16 MFMA products, f32 accumulate): an A/B arm, not the product default",
                                   "steps": ab_steps, "ms_per_step": 1e3 * dt_ab / ab_steps, "value": B * ab_steps / dt_ab,
                                   "unit": "clips/s", "roofline_stages": arm_stages}
        except Exception as ex:      # an A/B arm must never cost the run its headline measurement
            ab["error"] = repr(ex)
            sys.stderr.write("A/B leg failed: %r\n" % (ex,))
        finally:
            model.net.set_gemm_mode(0)
            _lib.Profiler.detach()   # a leg that raised between attach() and detach() must not leave this thread recording
    # ---- A/B of the backward schedules, same process, two alternating rounds each, all bit-identical (tests/test_net_gpu.py,
    # tests/test_fullsize_gpu.py).  gemm mode 1: the launches of rounds 1 - 3; mode 0 (the default): round 4's one-grid launches (a
    # layer's input-gradient + weight-gradient GEMM, slab sum beside the first convolution's weight gradient, the tail's post-kernels).
    # (Round 5's third schedule - weight-gradient work items beside the depthwise passes, a measured loss - left the library in round 6:
    # profiles/r05_wgrad_beside_dwbwd*.txt are its record, scripts/probes/wgrad_beside_dwbwd/ its code.)
    if world == 1 and not args.no_ab and gemm_mode == 0 and in_budget("ab_bwd_pair"):
        try:
            arms = {0: [], 1: []}
            for rnd in range(2):
                for mode in (1, 0):
                    model.net.set_gemm_mode(mode)
                    for i in range(4):
                        step(used + i)
                    dt_m, _ = timed_steps(used + 4, ab_steps)
                    used += 4 + ab_steps
                    arms[mode].append(1e3 * dt_m / ab_steps)
            ab["ab_bwd_pair"] = {"what": "all one-grid launches of gemm mode 0 (the default since round 4: input-gradient + weight-gradient GEMM of a layer, slab sum "
                                         "beside the first convolution's weight gradient, the tail's post-kernels) vs gemm mode 1 (the separate launches of "
                                         "rounds 1 - 3); best of two alternating rounds of %d steps" % ab_steps,
                                 "steps": ab_steps, "paired": {"ms_per_step": min(arms[0]), "value": B / min(arms[0]) * 1e3, "rounds_ms": arms[0]},
                                 "separate": {"ms_per_step": min(arms[1]), "value": B / min(arms[1]) * 1e3, "rounds_ms": arms[1]},
                                 "gain_us_per_step": 1e3 * (min(arms[1]) - min(arms[0])), "unit": "clips/s"}
        except Exception as ex:
            ab["ab_bwd_pair_error"] = repr(ex)
            sys.stderr.write("A/B backward-schedule leg failed: %r\n" % (ex,))
        finally:
            model.net.set_gemm_mode(0)
    # ---- configs[1]'s own A/B: "HIP STFT+mel vs raw-wave path".  The headline step produces BOTH arms of every batch (the
    # generator's 'mfcc_and_raw' output); here the same training step is timed with the generator switched between 'raw'
    # (augment only) and 'mfcc_and_raw' (augment + STFT/mel/DCT(80,60)) at run time, two alternating rounds each, same process.
    if world == 1 and not args.no_ab and in_budget("ab_features"):
        try:
            def step_any(i):
                X, y = enq.get()
                if isinstance(X, (list, tuple)):
                    X[0].wait()
                    X = X[1]
                model._train_step_async(X, y, ring[i])
            arms = {"raw": [], "mfcc_and_raw": []}
            for rnd in range(2):
                for rep in ("raw", "mfcc_and_raw"):
                    proc.output_representation = rep
                    for i in range(14):                 # the queue (depth 10) still holds batches of the other kind: train through them
                        step_any(used + i)
                    used += 14
                    barrier()
                    t0 = time.time()
                    for i in range(ab_steps):
                        step_any(used + i)
                    barrier()
                    arms[rep].append((time.time() - t0) / ab_steps)
                    used += ab_steps
            proc.output_representation = 'mfcc_and_raw'
            m_raw, m_both = 1e3 * min(arms["raw"]), 1e3 * min(arms["mfcc_and_raw"])
            ab["ab_features"] = {"what": "the same training step fed by the generator's 'raw' output (augment only) vs its "
                                         "'mfcc_and_raw' output (augment + STFT/mel/DCT(80,60) on the generator's low-priority stream); "
                                         "best of two alternating rounds of %d steps each" % ab_steps,
                                 "steps": ab_steps,
                                 "raw": {"ms_per_step": m_raw, "value": B / m_raw * 1e3, "rounds_ms": [1e3 * v for v in arms["raw"]]},
                                 "mfcc_and_raw": {"ms_per_step": m_both, "value": B / m_both * 1e3,
                                                  "rounds_ms": [1e3 * v for v in arms["mfcc_and_raw"]]},
                                 "stft_mel_cost_us_per_step": 1e3 * (m_both - m_raw), "unit": "clips/s"}
        except Exception as ex:
            ab["ab_features_error"] = repr(ex)
            sys.stderr.write("A/B features leg failed: %r\n" % (ex,))
    enq.stop()
    # ---- the surface north_star names: Model.fit_generator.  The headline loop above calls Model._train_step_async itself; here the
    # SAME generator and model are driven by model.fit_generator(gen, steps_per_epoch=100, epochs=1) - its own enqueuer thread, callback
    # dispatch, the L2-loss read every 16 steps (a host sync) and the per-epoch metric copy - against this file's loop over the same
    # number of steps from a fresh enqueuer, two alternating rounds.  Both arms are timed from "enqueuer not started" to "device idle".
    if world == 1 and not args.no_ab and in_budget("ab_fit_generator"):
        try:
            from speech_recognition_amd.device_array import DeviceArray

            def raw_after_features(g):
                # the headline step trains on `raw` once the STFT arm of the batch exists too (mfcc.wait()): the features' ready event is
                # recorded after raw's on the generator's stream, so it stands for both
                for (mfcc, raw), y in g:
                    yield DeviceArray(raw.tensor, mfcc.ready_event), y
            fit_steps = 100
            feed = raw_after_features(gen)
            ring_fg = torch.zeros((fit_steps, 4), dtype=torch.float32, device=device)
            arms = {"own_loop": [], "fit_generator": []}
            for rnd in range(2):
                barrier()
                t0 = time.time()
                e2 = GeneratorEnqueuer(feed, max_queue_size=10, device=device)
                e2.start()
                for i in range(fit_steps):
                    X, y = e2.get()
                    model._train_step_async(X, y, ring_fg[i])
                barrier()
                arms["own_loop"].append((time.time() - t0) / fit_steps)
                e2.stop()
                barrier()
                t0 = time.time()
                model.fit_generator(feed, steps_per_epoch=fit_steps, epochs=1, verbose=0, max_queue_size=10)
                barrier()
                arms["fit_generator"].append((time.time() - t0) / fit_steps)
            m_own, m_fit = 1e3 * min(arms["own_loop"]), 1e3 * min(arms["fit_generator"])
            ab["ab_fit_generator"] = {
                "what": "model.fit_generator(gen, steps_per_epoch=%d, epochs=1) - the surface train.py:69-71 calls - vs this file's own loop over "
                        "Model._train_step_async, same generator, same model, a fresh enqueuer (depth 10) each; best of two alternating rounds; "
                        "both timed from before the enqueuer starts until the device is idle" % fit_steps,
                "steps": fit_steps, "own_loop": {"ms_per_step": m_own, "value": B / m_own * 1e3, "rounds_ms": [1e3 * v for v in arms["own_loop"]]},
                "fit_generator": {"ms_per_step": m_fit, "value": B / m_fit * 1e3, "rounds_ms": [1e3 * v for v in arms["fit_generator"]]},
                "fit_generator_over_own_loop": m_fit / m_own, "unit": "clips/s"}
        except Exception as ex:
            ab["ab_fit_generator_error"] = repr(ex)
            sys.stderr.write("A/B fit_generator leg failed: %r\n" % (ex,))
    feature_err = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:                             # checker leg: the generator thread has stopped, this thread may draw from the processor
            feature_err = feature_error_vs_oracle(proc)
        except Exception as ex:
            feature_err = {"error": repr(ex)}
    if rank == 0 and stages:
        # the STFT stage ALONE (its in-situ time above is that of a low-priority stream filling the gaps of the training
        # stream): 20 back-to-back launches on one batch, HIP events on the launch stream
        try:
            torch.cuda.synchronize()
            (mf, raw), _ = next(gen)
            raw_t = raw.wait()
            st = proc._stream
            with torch.cuda.stream(st):
                for _ in range(3):
                    proc._features(raw_t, 0)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(st)
                for _ in range(20):
                    proc._features(raw_t, 0)
                e1.record(st)
            e1.synchronize()
            us = 1e3 * e0.elapsed_time(e1) / 20
            for e in stages:
                if e["family"] == "stft_mel":
                    # floors of the PRESENT kernel structure: HBM = algorithmic bytes at 8 TB/s;
                    # vector issue = the ISA census of a frame quad (435 FMA-class x 1.9 + 117 conversions / DPP / selects x 3.0 +
                    # 22 sqrt / log x 5.8 + 33 MFMA x 8 cycles of held issue = 1,776 cycles) x quads per SIMD at 2.1 GHz
                    quads = B * 25.0
                    e["stage_floor_us"] = {"hbm": e["algorithmic_bytes_per_launch"] / PEAK_HBM_GBS / 1e3,
                                           "vector_issue": 1776.0 * quads / 1024.0 / 2.1e3,
                                           "source": "profiles/r03_stft4_isa_per_quad_before_rewrite.txt x profiles/r02_probe_valu_rates.txt"}
                    e["alone_avg_launch_us"] = us
                    e["alone_frac_hbm"] = e["algorithmic_bytes_per_launch"] / (us * 1e-6) / 1e9 / PEAK_HBM_GBS
                    e["alone_frac_flops"] = e["algorithmic_flops_per_launch"] / (us * 1e-6) / 1e12 / PEAK_F32_MFMA_TFLOPS
            # the augment stage ALONE, the same way (its in-situ 0.44 of HBM is measured beside the training step on a low-priority
            # stream: this says whether the kernel or the contention is short of the roof): the last batch's own draw parameters,
            # 20 back-to-back launches
            di, df, do = proc._keep
            Bk = int(di.numel() // 2)
            bank = proc.bank
            fn = "kws_augment_i16" if bank.clips.dtype == torch.int16 else "kws_augment_f32"
            out_t = torch.empty((Bk, 16000), dtype=torch.float32, device=device)

            def aug():
                _lib.call(fn, _lib.ptr(bank.clips), bank.n_clips, 16000, _lib.ptr(di[:Bk]), _lib.ptr(df[:Bk]), _lib.ptr(di[Bk:]),
                          _lib.ptr(bank.noise), 0 if bank.noise is None else bank.noise.numel(),
                          _lib.ptr(do) if bank.noise is not None else None, _lib.ptr(df[Bk:]), _lib.ptr(out_t), Bk, _lib.stream_ptr(st))
            with torch.cuda.stream(st):
                for _ in range(3):
                    aug()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(st)
                for _ in range(20):
                    aug()
                e1.record(st)
            e1.synchronize()
            us_a = 1e3 * e0.elapsed_time(e1) / 20
            for e in stages:
                if e["family"] == "augment":
                    e["alone_avg_launch_us"] = us_a
                    e["alone_frac_hbm"] = e["algorithmic_bytes_per_launch"] / (us_a * 1e-6) / 1e9 / PEAK_HBM_GBS
        except Exception as ex:
            sys.stderr.write("standalone STFT / augment timing skipped: %r\n" % (ex,))

    # ---- the other single-GPU BASELINE configurations on the record (N = 1 only; < 2 s each): configs[2] = C3, the 32-class
    # conv_1d_log_mfcc net at batch 2048 (freeze_graph_32_classes.py:55-69), configs[4] = C5 on this one GPU, TTA inference
    # (make_submission.py:120-146) at batch 4096.  Their rocprofv3 summaries: profiles/r03_kernel_stats_c3.csv / _c5.csv.
    configs = None
    if world == 1 and not args.no_configs and not in_budget("configs"):
        configs = {"skipped": "budget"}
    elif world == 1 and not args.no_configs:
        configs = {}
        try:
            sys.path.insert(0, os.path.join(ROOT, "scripts"))
            import bench_configs
            proc.close()                              # the headline leg's generator stream and clip bank are done
            del proc, spec, enq, gen
            torch.cuda.empty_cache()
            c3 = bench_configs.c3(steps=30, warm=5, profile_steps=3)
            kern = c3.pop("kernels", None) or {}
            fams = sorted([f for f in kern if kern[f]["ms"] > 0], key=lambda f: -kern[f]["ms"])
            c3["kernel_ms_per_step"] = {f: kern[f]["ms"] / 3 for f in fams[:10]}
            c3["launches_per_step"] = sum(kern[f]["count"] for f in kern) / 3.0
            # every family of the C3 step against both roofs, with the PMC traffic and the rocprofv3 durations of scripts/profile_c3.sh
            # (profiles/r0N_pmc_traffic_c3.json, r0N_kernel_stats_c3.csv: the DEFAULT schedule only) while the kernels' sources are unchanged
            pmc3, roc3 = load_pmc_traffic("r*_pmc_traffic_c3.json"), load_rocprof_stats("r*_kernel_stats_c3.csv")
            c3_stages = []
            for family, kernel, bound, what in C3_ROOFLINE_KERNELS:
                e = roofline_entry(kern, family, kernel, bound, what, pmc3, roc3)
                if e is not None:
                    e["ms_per_step"] = kern[family]["ms"] / 3
                    c3_stages.append(e)
            gemm3 = [e for e in c3_stages if e["family"] in ("gemm_bwd_pair", "gemm_nn")]
            c3["roofline"] = max(gemm3, key=lambda e: e["ms_per_step"]) if gemm3 else None
            c3["roofline_stages"] = [e for e in c3_stages if e is not c3["roofline"]]
            gflop = sum(kern[f]["flops"] for f in kern if f.startswith("gemm")) / 3.0
            c3["end_to_end"] = {"gemm_flops_per_step": gflop, "frac_of_f32_mfma_peak": gflop / (c3["ms_per_step"] * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS,
                                "algorithmic_bytes_per_step": sum(kern[f]["bytes"] for f in kern) / 3.0,
                                "frac_of_hbm_peak": sum(kern[f]["bytes"] for f in kern) / 3.0 / (c3["ms_per_step"] * 1e-3) / 1e9 / PEAK_HBM_GBS}
            configs["C3"] = c3
            c5 = bench_configs.c5(speed_tta=True, n=10, warm=3)   # x3 AND the six-term speed TTA BASELINE configs[4] names
            k5 = c5.pop("kernels", None) or {}
            if k5:      # plain inference (one forward of 4096 clips): the dominant family + the end-to-end MFMA fraction
                n5 = float(c5.get("profiled_batches", 1))
                roc5 = load_rocprof_stats("r*_kernel_stats_c5.csv")
                e5 = roofline_entry(k5, "gemm_nn", "gemm_nn_ws_kernel", "mfma", "pointwise 1x1 convolutions, forward (plain inference, batch 4096)", None, roc5)
                gf5 = sum(k5[f]["flops"] for f in k5 if f.startswith("gemm") or f.startswith("conv1")) / n5
                c5["roofline"] = e5
                c5["roofline_stages"] = [e for e in (roofline_entry(k5, f, kn, b, w, None, roc5) for f, kn, b, w in C5_ROOFLINE_KERNELS) if e is not None]
                c5["end_to_end"] = {"gemm_flops_per_batch": gf5,
                                    "frac_of_f32_mfma_peak": gf5 / (c5["plain_inference_ms_per_batch"] * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS}
            configs["C5"] = c5
        except Exception as ex:
            configs["error"] = repr(ex)
            sys.stderr.write("configs legs failed: %r\n" % (ex,))

    anchor = None
    if rank == 0 and world == 1 and not args.no_configs and in_budget("epoch_anchor"):
        try:
            anchor = epoch_anchor(device)
        except Exception as ex:
            anchor = {"error": repr(ex)}
            sys.stderr.write("epoch anchor leg failed: %r\n" % (ex,))

    out = None
    if rank == 0:
        out = {
            "metric": "1s 16kHz clips/sec training throughput",
            "value": clips / dt, "unit": "clips/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak",
            # BASELINE.md publishes no clips/s; its one comparable anchor is exp-195's logged wall time per epoch (193.7 s), re-run here
            # through fit_generator with the same epoch shape: vs_baseline = that ratio of epochs per second
            "vs_baseline": (anchor or {}).get("epochs_per_s_ratio"),
            "baseline": "logs_195 wall time per epoch (193.7 s: 95 x 384 training clips + 4,224 validation clips), host apple2, GPU unknown, real "
                        "wavs - vs epoch_anchor.seconds_per_epoch on this MI355X, synthetic HBM-resident bank; a ratio of epochs/s, not of `value`",
            "epoch_anchor": anchor,
            "dtype": {0: "f32", 1: "f32", 3: "f32", 2: "f32 (pointwise GEMMs as scaled 2-way fp16 splits, f32 accumulate)"}[gemm_mode],
            "data": "synthetic",
            "config": {"workload": "configs[1]: 12-class conv_1d_time_sliced_with_attention, batch %d/GPU synthetic "
                                   "16000-sample fp32 clips, sampler+augment+STFT/mel(80,60)+raw fwd/bwd+RMSprop" % B,
                       "global_batch": B * world, "parallelism": "dp%d" % world,
                       "clip_bank": args.bank},
            "rccl_ranks": rccl_ranks,
            "preflight": preflight,
            "collective_backend": (dist.get_backend() if dist else None),
            "per_rank_ms": per_rank_ms,
            "scaling_vs_n1": (clips / dt) / (world * args.n1_value) if args.n1_value else None,
            "sampler_seeds": sampler_seeds,
            "allreduce_only": exchange,
            "ab_allreduce_split": ab.get("ab_allreduce_split"),
            "train_loss_first_last": [float(ms[0, 0] / B), float(ms[-1, 0] / B)],
            "train_acc_last": float(ms[-1, 1] / B),
            "roofline": roof,
            "roofline_stages": [e for e in stages if e is not roof],
            "ab_gemm_f16x2": ab.get("ab_gemm_f16x2"),
            "ab_features": ab.get("ab_features"),
            "ab_fit_generator": ab.get("ab_fit_generator"),
            "ab_fit_generator_error": ab.get("ab_fit_generator_error"),
            "ab_bwd_pair": ab.get("ab_bwd_pair"),
            "ab_bwd_pair_error": ab.get("ab_bwd_pair_error"),
            "stft_mel_error": feature_err,
            "ab_error": ab.get("error"),
            "configs": configs,
            "kernels": prof,
            "skipped_legs": skipped,
        }
    if dist:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        # the host-core baseline of the same run: after the timed region and after the process group is gone, so no
        # rank waits in a collective for it (under bench.py's own launcher the parent process measures it instead).  The contract's
        # object: it runs whatever the wall budget says, and BEFORE the longest checker leg
        if not args.no_cpu_baseline and not os.environ.get("KWS_BENCH_CHILD"):
            try:
                out["cpu_baseline"] = cpu_baselines()
            except Exception as e:
                out["cpu_baseline"] = {"error": repr(e)}
        if not args.no_val_acc and world == 1:
            if in_budget("val_acc_parity"):
                try:
                    sys.path.insert(0, os.path.join(ROOT, "scripts"))
                    import val_acc_parity
                    # ONE sampler seed here (the CPU twin is 50 - 75 s per seed on the box's host cores, the longest leg of the run by far); the
                    # two-seed form is tests/test_val_acc_gpu.py and the script's own default
                    out.update(val_acc_parity.run(device, quiet=True, seeds=(val_acc_parity.SEEDS[0],), negative_controls=("dw_flip",)))
                except Exception as e:
                    out["val_acc_error"] = repr(e)
            else:
                out["val_acc_parity"] = {"skipped": "budget"}
        out["skipped_legs"] = skipped
        out["bench_wall_s"] = time.time() - t_start
        print(json.dumps(out), file=json_out, flush=True)


if __name__ == "__main__":
    main()
