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


def build_synthetic(device, n_bank, seed, L=16000):
    """SURVEY 8d synthetic inputs: x = 0.0774*N(0,1) clipped to [-1,1] + 0.05 sin(2 pi f_c t), f_c = 200(1+c) Hz;
    6 x 60 s noise recordings; label mix silence 13 % / unknown 60 % (train.py:40-45); a 'pseudo' partition."""
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
        f = 200.0 * (1.0 + cls_t[s:e].float())
        x += 0.05 * torch.sin(2.0 * math.pi * f[:, None] * t[None, :])
        bank[s:e] = x.clamp_(-1.0, 1.0)
    rng = np.random.RandomState(59185)
    noise = [(rng.randn(960000) * 0.1).astype(np.float32) for _ in range(6)]
    cb = ClipBank(bank, noise, device)
    n_pseudo = n_bank // 8
    train_rows = [r for r in range(n_wanted - n_pseudo // 2)]
    pseudo_rows = [r for r in range(n_wanted - n_pseudo // 2, n_wanted)]
    unk_rows = list(range(n_wanted, n_bank))
    n_sil = int(math.ceil(len(train_rows) * 13.0 / 100))
    n_unk = min(int(math.ceil(len(train_rows) * 60.0 / 100)), len(unk_rows) - n_pseudo // 2)
    index = {
        'training': [(r, word_of_row[r]) for r in train_rows] + [(0, SILENCE_LABEL)] * n_sil +
                    [(r, word_of_row[r]) for r in unk_rows[:n_unk]],
        'pseudo': [(r, word_of_row[r]) for r in pseudo_rows] + [(r, word_of_row[r]) for r in unk_rows[n_unk:n_unk + n_pseudo // 2]],
        'validation': [(r, word_of_row[r]) for r in range(0, min(4096, n_bank))],
        'testing': [],
    }
    return {'bank': cb, 'index': index}


def cpu_baseline(seconds_budget=20.0):
    """The oracle ("port") timed on this box's host cores: reference-style per-clip generator (one clip
    per call, float64 batch buffer) feeding full train steps at B=64 (BASELINE configs[0])."""
    from oracle import features as OF
    from oracle.net import TimeSlicedAttentionNet
    try:
        from threadpoolctl import threadpool_info
        threads = max([p.get('num_threads', 1) for p in threadpool_info()] or [1])
    except Exception:
        threads = os.cpu_count() or 1
    B = 64
    rng = np.random.RandomState(0)
    bank = (rng.randn(256, 16000) * 0.0774).astype(np.float32)
    noise = (rng.randn(960000) * 0.1).astype(np.float32)
    tables = OF.tables_path_b(480, 80, 60)
    net = TimeSlicedAttentionNet(num_classes=12, dtype=np.float32)
    net.init_optimizer('rmsprop')
    steps, t0 = 0, time.time()
    while True:
        data = np.zeros((B, 16000))
        feats = np.zeros((B, 98 * 60))
        lab = rng.randint(0, 12, B)
        for i in range(B):   # per-clip loop, like input_data.py:457-536
            clip = OF.augment(bank[rng.randint(256)], 1.0 + rng.uniform(-0.15, 0.15), rng.randint(-500, 1),
                              noise[rng.randint(0, 960000 - 16000):][:16000], rng.uniform(0, 0.15))
            data[i, :] = clip
            feats[i, :] = OF.features(clip, tables, 160, dtype=np.float32).reshape(-1)
        net.train_step(data.astype(np.float32), np.eye(12, dtype=np.float32)[lab], 1e-3, seed=1, step=steps)
        steps += 1
        if time.time() - t0 > seconds_budget or steps >= 8:
            break
    dt = time.time() - t0
    return {"value": B * steps / dt, "unit": "clips/s", "cores": int(threads), "kind": "port",
            "sample": "%d train steps at batch 64 (per-clip augment + STFT/mel/DCT + f32 NumPy fwd/bwd/RMSprop), %.1f s"
                      % (steps, dt)}


def main():
    # stdout carries exactly ONE line (the JSON result); everything else the pipeline prints on the way
    # (data_gen's per-epoch "[Ep:...]" line mirrors the reference and goes to stdout) is sent to stderr
    json_out = sys.stdout
    sys.stdout = sys.stderr
    if os.environ.get("KWS_BENCH_TRACE"):        # debugging aid: dump every thread's stack after N seconds and exit
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["KWS_BENCH_TRACE"]), exit=True)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=15)
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--bank", type=int, default=65536)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--profile-steps", type=int, default=5)
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("KWS_BENCH_ONE_DEVICE"):   # test hook: N ranks on ONE GPU over gloo (1-GPU boxes cannot run RCCL x N)
        local_rank = 0
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: there is no CPU fallback of the product path")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if os.environ.get("KWS_BENCH_ONE_DEVICE"):
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
    assert world == args.gpus or world == 1, "launch with torch.distributed.run --nproc-per-node %d" % args.gpus

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
    ring = torch.zeros((args.warmup + args.steps + args.profile_steps + 8, 4), dtype=torch.float32, device=device)
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

    for i in range(args.warmup):
        step(i)
    barrier()
    t0 = time.time()
    for i in range(args.steps):
        step(args.warmup + i)
    barrier()
    dt = time.time() - t0
    if dist:
        tmax = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    clips = B * world * args.steps
    ms = ring[args.warmup:args.warmup + args.steps].cpu().numpy()

    # ---- per-kernel durations of the same step, HIP events on the launch stream -------------------
    prof = None
    roof = None
    if args.profile_steps > 0:
        # EVERY rank runs the profiled steps (each step contains the gradient all-reduce: a rank that skipped them
        # would leave the others waiting in the collective); only rank 0 records and reports the kernel times
        lib = _lib.load()
        if rank == 0:
            lib.kws_profile_enable(1)
        for i in range(args.profile_steps):
            step(args.warmup + args.steps + i)
        barrier()
    if rank == 0 and args.profile_steps > 0:
        prof = _lib.profile_collect()
        lib.kws_profile_enable(0)
        k = prof.get("gemm_nn")
        if True:
            if k and k["ms"] > 0:
                ach = k["flops"] / (k["ms"] * 1e-3) / 1e12
                traffic = None   # HBM bytes per launch from the separate rocprofv3 --pmc passes (profiles/)
                try:
                    with open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")) as f:
                        traffic = json.load(f)["kernels"]["gemm_nn_ws_kernel"]["hbm_bytes_per_launch"]
                except Exception:
                    pass
                roof = {"kernel": "gemm_nn_ws_kernel (f32 MFMA: pointwise fwd + dgrad; the first convolution's forward "
                                  "runs conv1_fwd_kernel and is counted in the same family)",
                        "bound": "mfma", "achieved": ach, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                        "frac": ach / PEAK_F32_MFMA_TFLOPS, "traffic": traffic,
                        "algorithmic_bytes_per_launch": k["bytes"] / max(k["count"], 1),
                        "launches": k["count"], "avg_launch_us": 1e3 * k["ms"] / max(k["count"], 1)}
    enq.stop()

    if rank == 0:
        out = {
            "metric": "1s 16kHz clips/sec training throughput",
            "value": clips / dt, "unit": "clips/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "configs[1]: 12-class conv_1d_time_sliced_with_attention, batch %d/GPU synthetic "
                                   "16000-sample fp32 clips, sampler+augment+STFT/mel(80,60)+raw fwd/bwd+RMSprop" % B,
                       "global_batch": B * world, "parallelism": "dp%d" % world,
                       "clip_bank": args.bank},
            "train_loss_first_last": [float(ms[0, 0] / B), float(ms[-1, 0] / B)],
            "train_acc_last": float(ms[-1, 1] / B),
            "roofline": roof,
            "kernels": prof,
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline()
        elif not args.no_cpu_baseline:
            out["cpu_baseline"] = None
        print(json.dumps(out), file=json_out, flush=True)
    if dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
