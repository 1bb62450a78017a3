"""SURVEY 8(b), threading clause: Keras calls next(train_gen) from its enqueuer thread while the main thread trains
(reference train.py:69-71, fit_generator's GeneratorEnqueuer), so the feature / augment entry points must be callable from
one host thread on one stream WHILE another host thread runs the training program on another stream - no shared mutable
state in the library, kws_last_error() per thread.  Two Python threads (ctypes releases the GIL inside every call), two HIP
streams, every iteration compared bit for bit with what the same calls return on one thread."""
import ctypes
import threading

import numpy as np
import pytest
import torch

from oracle import features as OF
from speech_recognition_amd import _lib
from speech_recognition_amd.net import DeviceNet

pytestmark = pytest.mark.gpu


def _plan(tables, frame_step, n_mel, n_out):
    lib = _lib.load()
    win = np.ascontiguousarray(tables["window"], dtype=np.float32)
    mel = np.ascontiguousarray(tables["mel"], dtype=np.float32)
    dct = np.ascontiguousarray(tables["dct"], dtype=np.float32)
    plan = ctypes.c_void_p()
    _lib.check(lib.kws_stft_plan_create(len(win), frame_step, 512, n_mel, n_out, win.ctypes.data_as(ctypes.c_void_p),
                                        mel.ctypes.data_as(ctypes.c_void_p), dct.ctypes.data_as(ctypes.c_void_p),
                                        tables["log_offset"], tables["log_floor"], ctypes.byref(plan)), "plan_create")
    return plan


def test_generator_thread_and_training_thread_on_two_streams():
    lib = _lib.load()
    dev = torch.device("cuda", 0)
    rng = np.random.RandomState(5)
    # ---- generator side: augment (a2) + STFT/mel/DCT (a3 - a4) of 96 clips
    n_clips, L, Bg = 64, 16000, 96
    bank = torch.from_numpy((rng.randn(n_clips, L) * 0.08).astype(np.float32)).to(dev)
    noise_h = (rng.randn(6 * 16000) * 0.1).astype(np.float32)
    noise = torch.from_numpy(noise_h).to(dev)
    idx_h = rng.randint(0, n_clips, Bg).astype(np.int32)
    fg_h = (1 + rng.uniform(-0.15, 0.15, Bg)).astype(np.float32)
    shift_h = rng.randint(-500, 1, Bg).astype(np.int32)
    off_h = rng.randint(0, len(noise_h) - L, Bg).astype(np.int64)
    bgv_h = rng.uniform(0, 0.15, Bg).astype(np.float32)
    idx, fg, shift, off, bgv = (torch.from_numpy(a).to(dev) for a in (idx_h, fg_h, shift_h, off_h, bgv_h))
    tables = OF.tables_path_b(480, 80, 60)
    plan = _plan(tables, 160, 80, 60)
    F = lib.kws_stft_num_frames(plan, L)

    def generate(stream, raw, feat):
        _lib.call("kws_augment_f32", _lib.ptr(bank), n_clips, L, _lib.ptr(idx), _lib.ptr(fg), _lib.ptr(shift), _lib.ptr(noise),
                  noise.numel(), _lib.ptr(off), _lib.ptr(bgv), _lib.ptr(raw), Bg, _lib.stream_ptr(stream))
        _lib.call("kws_stft_mel_f32", plan, _lib.ptr(raw), Bg, L, _lib.ptr(feat), 0, _lib.stream_ptr(stream))

    # ---- training side: the raw-waveform net's whole forward + backward at batch 48
    Bt = 48
    net = DeviceNet(_lib.KWS_NET_TS_ATTENTION, 12)
    net.initialize(seed=3)
    w0, s0 = net.params.clone(), net.state.clone()
    x = torch.from_numpy((rng.randn(Bt, 16000) * 0.0774).astype(np.float32)).to(dev)
    y = torch.eye(12, device=dev)[torch.from_numpy(rng.randint(0, 12, Bt)).to(dev)].contiguous()

    def train(stream, probs):
        with torch.cuda.stream(stream):
            net.params.copy_(w0)
            net.state.copy_(s0)
        return net.train_fwd_bwd(x, y, seed=17, step=2, probs=probs, stream=stream)

    # ---- the single-thread results (one stream, nothing beside them), and the oracle bars the kernels' own tests hold them to
    torch.cuda.synchronize()            # the operands above were written on the default stream; torch's side streams do not wait for it
    s_ref = torch.cuda.Stream()
    raw_ref = torch.full((Bg, L), float("nan"), device=dev)
    feat_ref = torch.full((Bg, F, 60), float("nan"), device=dev)
    generate(s_ref, raw_ref, feat_ref)
    p_ref = train(s_ref, torch.full((Bt, 12), float("nan"), device=dev))
    s_ref.synchronize()
    p_ref, g_ref, st_ref, m_ref = p_ref.clone(), net.grads.clone(), net.state.clone(), net.metrics.clone()
    torch.cuda.synchronize()
    aug64 = OF.augment_batch(bank.cpu().numpy(), idx_h, fg_h, shift_h, noise_h, off_h, bgv_h, dtype=np.float32)
    assert np.array_equal(raw_ref.cpu().numpy(), aug64)
    feat64 = OF.features(aug64[:8].astype(np.float64), tables, 160, dtype=np.float64)
    assert np.abs(feat_ref[:8].cpu().numpy() - feat64).max() < 1.1e-4
    assert float(g_ref.abs().max()) > 0 and bool(torch.isfinite(g_ref).all())

    # ---- two threads, two streams
    N_GEN, N_TRAIN = 40, 12
    failures = []
    gate = threading.Barrier(2)
    err_set, err_checked = threading.Event(), threading.Event()
    seen = {}

    def last_error():
        return lib.kws_last_error()      # (restype c_char_p: bytes; the buffer is thread-local in the library)

    def generator_thread():
        try:
            torch.cuda.set_device(dev)
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):  # (allocated and filled on the stream that uses them)
                raws = [torch.full((Bg, L), float("nan"), device=dev) for _ in range(2)]
                feats = [torch.full((Bg, F, 60), float("nan"), device=dev) for _ in range(2)]
            seen["gen_before"] = last_error()
            gate.wait(30)
            for i in range(N_GEN):
                generate(st, raws[i % 2], feats[i % 2])
                st.synchronize()
                with torch.cuda.stream(st):
                    same = torch.equal(raws[i % 2], raw_ref) and torch.equal(feats[i % 2], feat_ref)
                    raws[i % 2].fill_(float("nan")); feats[i % 2].fill_(float("nan"))
                if not same:
                    failures.append("generator iteration %d differs from the single-thread result" % i)
                if i == N_GEN // 2:
                    # a call that must fail, on THIS thread only: a NULL clip bank
                    with pytest.raises(_lib.KwsError):
                        _lib.call("kws_augment_f32", None, n_clips, L, _lib.ptr(idx), _lib.ptr(fg), _lib.ptr(shift), _lib.ptr(noise),
                                  noise.numel(), _lib.ptr(off), _lib.ptr(bgv), _lib.ptr(raws[0]), Bg, _lib.stream_ptr(st))
                    seen["gen_after_failure"] = last_error()
                    err_set.set()
                    err_checked.wait(30)
        except Exception as ex:          # noqa: BLE001 - reported by the main thread
            failures.append("generator thread: %r" % (ex,))
            err_set.set()

    def training_thread():
        try:
            torch.cuda.set_device(dev)
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                probs = torch.full((Bt, 12), float("nan"), device=dev)
            gate.wait(30)
            for i in range(N_TRAIN):
                p = train(st, probs)
                st.synchronize()
                with torch.cuda.stream(st):
                    same = (torch.equal(p, p_ref) and torch.equal(net.grads, g_ref) and torch.equal(net.state, st_ref) and
                            torch.equal(net.metrics, m_ref))
                if not same:
                    failures.append("training iteration %d differs from the single-thread result" % i)
                if i == 2:
                    err_set.wait(30)
                    seen["train_while_gen_failed"] = last_error()     # the other thread's message must not show here
                    err_checked.set()
            # and the other way round: this thread fails (loss_batch < B), its message stays here
            with pytest.raises(_lib.KwsError):
                net.train_fwd_bwd(x, y, seed=17, step=2, loss_batch=Bt - 1, stream=st)
            seen["train_after_failure"] = last_error()
        except Exception as ex:          # noqa: BLE001
            failures.append("training thread: %r" % (ex,))
            err_checked.set()

    tg, tt = threading.Thread(target=generator_thread), threading.Thread(target=training_thread)
    tg.start(); tt.start()
    tg.join(120); tt.join(120)
    assert not tg.is_alive() and not tt.is_alive(), "a thread hung"
    assert not failures, failures
    assert seen["gen_before"] == b"" and seen["train_while_gen_failed"] == b"", seen
    assert b"augment" in seen["gen_after_failure"] and b"loss_batch" in seen["train_after_failure"], seen
    _lib.check(lib.kws_stft_plan_destroy(plan), "plan_destroy")
