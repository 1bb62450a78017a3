"""The other single-GPU BASELINE.json configurations, measured on one MI355X (bench.py is the contract benchmark for
configs[1] and calls these as its `configs` legs; `python scripts/bench_configs.py` prints them alone, and
scripts/prof_c3.py / prof_c5.py are the rocprofv3 targets):

  C3  configs[2]: 32-class conv_1d_log_mfcc net (model.py:1400-1479; the net freeze_graph_32_classes.py:55-69 freezes) on
      40 x 98 log-mel features, batch 2048: STFT/mel(40,40) of the raw clips + forward/backward + RMSprop (6e-4),
      inputs resident in HBM;
  C5  configs[4] on one GPU: TTA inference (identity + 1.2x volume + 1500-sample roll, make_submission.py:120-146) of the
      12-class raw-waveform net, batches of 4096 clips; plain inference for comparison; and the six-term speed
      TTA with the slow clips stretched on the device.

Every function returns a dict (one JSON object per configuration)."""
import ctypes
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from speech_recognition_amd import _lib  # noqa: E402
from speech_recognition_amd.features import path_b_tables  # noqa: E402
from speech_recognition_amd.keras_api import Model, RMSprop  # noqa: E402
from speech_recognition_amd.net import DeviceNet  # noqa: E402
from speech_recognition_amd.tta import predict_tta, time_stretch  # noqa: E402

ARM = {0: "f32 MFMA (product default)", 2: "fp16 x 2 split (A/B arm)"}


def timed(fn, warm, n):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


def clips(B, seed):
    g = torch.Generator(device="cuda")
    g.manual_seed(seed)
    return (torch.randn((B, 16000), generator=g, device="cuda") * 0.0774).clamp_(-1, 1).contiguous()


def c3(steps=30, warm=5, profile_steps=5, ab=True):
    B = 2048
    lib = _lib.load()
    t = path_b_tables(480, 40, 40)
    plan = ctypes.c_void_p()
    _lib.check(lib.kws_stft_plan_create(480, 160, 512, 40, 40, t['window'].ctypes.data_as(ctypes.c_void_p),
                                        t['mel'].ctypes.data_as(ctypes.c_void_p), t['dct'].ctypes.data_as(ctypes.c_void_p),
                                        1e-6, 0.0, ctypes.byref(plan)), "plan")
    net = DeviceNet(_lib.KWS_NET_LOG_MFCC, 32, input_size=98 * 40, spectrogram_length=98, num_features=40)
    net.initialize(seed=3)
    model = Model(net, RMSprop(lr=6e-4), loss='cce')
    x = clips(B, 1)
    y = torch.eye(32, device="cuda")[torch.randint(0, 32, (B,), device="cuda")].contiguous()
    feats = torch.empty((B, 98 * 40), device="cuda")
    row = torch.zeros(4, device="cuda")

    def step():
        _lib.call("kws_stft_mel_f32", plan, _lib.ptr(x), B, 16000, _lib.ptr(feats), 0, _lib.stream_ptr())
        model._train_step_async(feats, y, row)

    ms = timed(step, warm, steps)
    ms_sep = None
    if ab:                           # A/B reference: the input- / weight-gradient GEMMs as separate launches (rounds 1 - 3);
        net.set_gemm_mode(1)         # never inside a profiled process (ab=False there: the stats must be ONE schedule)
        ms_sep = timed(step, 3, steps)
        net.set_gemm_mode(0)
    out = {"config": "C3 (configs[2]): 32-class conv_1d_log_mfcc, log-mel 40x98 from raw clips, batch 2048, "
                     "STFT/mel + fwd + bwd + RMSprop",
           "batch": B, "steps": steps, "ms_per_step": ms, "clips_per_s": B / ms * 1e3, "n_gpus": 1, "dtype": "f32",
           "gemm_arm": ARM[net.gemm_mode], "data": "synthetic",
           }
    if ab:
        out["ab_bwd_pair"] = {"paired_ms_per_step": ms, "separate_ms_per_step": ms_sep, "gain_us_per_step": 1e3 * (ms_sep - ms)}
    if profile_steps > 0:        # per-family HIP-event times of the same step (the caller turns the largest into a roofline object)
        prof = _lib.Profiler()
        prof.attach()
        for _ in range(profile_steps):
            step()
        torch.cuda.synchronize()
        out["kernels"] = prof.collect()
        prof.detach()
        prof.close()
    lib.kws_stft_plan_destroy(plan)
    return out


def c5(speed_tta=True, n=10, warm=3):
    B = 4096
    net = DeviceNet(_lib.KWS_NET_TS_ATTENTION, 12)
    net.initialize(seed=3)
    model = Model(net, RMSprop())
    x = clips(B, 2)
    ms_plain = timed(lambda: net.predict(x), warm, n)
    ms_tta = timed(lambda: predict_tta(model, x), warm, n)
    out = {"config": "C5 (configs[4] on one GPU): 12-class raw-waveform net, TTA inference x3 (identity, 1.2x, roll 1500), "
                     "batch 4096",
           "batch": B, "ms_per_batch": ms_tta, "clips_per_s": B / ms_tta * 1e3, "plain_inference_ms_per_batch": ms_plain,
           "plain_inference_clips_per_s": B / ms_plain * 1e3, "n_gpus": 1, "dtype": "f32", "gemm_arm": ARM[net.gemm_mode],
           "data": "synthetic"}
    # per-family HIP-event times of PLAIN inference (the caller builds C5's roofline object from them)
    prof = _lib.Profiler()
    prof.attach()
    for _ in range(3):
        net.predict(x)
    torch.cuda.synchronize()
    out["kernels"] = prof.collect()
    out["profiled_batches"] = 3
    prof.detach()
    prof.close()
    if speed_tta:
        # make_submission.py use_speed_tta: three more passes over the 0.9x time-stretched clips, stretched on the
        # device inside the timed region (the reference reads them from the offline set of create_tta_set.py)
        ms_tta6 = timed(lambda: predict_tta(model, x, use_speed_tta=True), warm, n)
        ms_stretch = timed(lambda: time_stretch(x, 0.9), warm, n)
        # the stretch kernel against both roofs: 64 kB in + 64 kB out per clip (f32 bank), ~8 MFLOP per clip (68 real
        # 2048-point FFTs as 1024-point complex radix-4 transforms + phase vocoder)
        by, fl = 128000.0 * B, 8.0e6 * B
        out["speed_tta"] = {"what": "x6 (identity, 1.2x, roll 1500, slow, clip(1.1 slow), 0.9 slow; the sum divided by 10 as "
                                    "make_submission.py:137-140 does), phase-vocoder stretch on the device",
                            "ms_per_batch": ms_tta6, "clips_per_s": B / ms_tta6 * 1e3,
                            "stretch_ms_per_batch": ms_stretch, "stretch_clips_per_s": B / ms_stretch * 1e3,
                            "stretch_roofline": {"kernel": "stretch_kernel", "bound": "hbm",
                                                 "achieved": by / (ms_stretch * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                                                 "frac": by / (ms_stretch * 1e-3) / 1e9 / 8000.0,
                                                 "frac_flops": fl / (ms_stretch * 1e-3) / 1e12 / 157.3,
                                                 "algorithmic_bytes_per_launch": by, "algorithmic_flops_per_launch": fl,
                                                 "avg_launch_us": ms_stretch * 1e3,
                                                 "note": "latency / VALU bound (340 barrier-to-barrier passes per clip); timed by HIP "
                                                         "events on the launch stream, %d launches" % n}}
    return out


if __name__ == "__main__":
    print(json.dumps(c3()))
    print(json.dumps(c5()))
