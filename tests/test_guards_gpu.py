"""Guard-band tests (SURVEY 5 build note; GPU AddressSanitizer is not available on this pool): every output of the hot
kernels is a window inside a larger allocation whose margins hold a sentinel bit pattern.  After the launch the margins
must be untouched - a kernel that writes one element past its output (ragged last tile, a frame quad past the clip's
last frame, a row past M) fails here even when the values inside the window are right."""
import ctypes

import numpy as np
import pytest
import torch

from speech_recognition_amd import _lib

pytestmark = pytest.mark.gpu

GUARD = 4096                       # floats on each side
SENT = 0x7FC0DEAD                  # a quiet-NaN payload no kernel produces


class Guarded(object):
    def __init__(self, shape, dtype=torch.float32):
        n = int(np.prod(shape))
        self.buf = torch.empty(n + 2 * GUARD, dtype=torch.int32, device="cuda")
        self.buf.fill_(SENT)
        self.view = self.buf[GUARD:GUARD + n].view(dtype).view(*shape)
        assert dtype in (torch.float32, torch.int32)

    def check(self, what):
        torch.cuda.synchronize()
        lo, hi = self.buf[:GUARD], self.buf[-GUARD:]
        assert bool((lo == SENT).all()) and bool((hi == SENT).all()), "%s wrote outside its output" % what
        assert not bool((self.view.view(torch.int32) == SENT).any()), "%s left output elements unwritten" % what


def S():
    return _lib.stream_ptr()


@pytest.mark.parametrize("B,L", [(1, 16000), (37, 16000), (3, 1002)])
def test_augment_stays_inside(B, L):
    g = torch.Generator(device="cuda")
    g.manual_seed(B)
    bank = torch.randn((64, L), generator=g, device="cuda")
    idx = torch.randint(0, 64, (B,), generator=g, device="cuda", dtype=torch.int32)
    shift = torch.randint(-L - 3, L + 4, (B,), generator=g, device="cuda", dtype=torch.int32)
    fg = torch.rand(B, generator=g, device="cuda")
    bgv = torch.rand(B, generator=g, device="cuda")
    noise = torch.randn(3 * L, generator=g, device="cuda")
    noff = torch.randint(0, 2 * L, (B,), generator=g, device="cuda", dtype=torch.int64)
    out = Guarded((B, L))
    _lib.call("kws_augment_f32", _lib.ptr(bank), 64, L, _lib.ptr(idx), _lib.ptr(fg), _lib.ptr(shift), _lib.ptr(noise),
              noise.numel(), _lib.ptr(noff), _lib.ptr(bgv), _lib.ptr(out.view), B, S())
    out.check("kws_augment_f32")


@pytest.mark.parametrize("B,n_mel,n_out,win,step,kind", [(1, 80, 60, 480, 160, 0), (5, 80, 60, 480, 160, 0), (3, 40, 40, 480, 160, 0),
                                                         (7, 40, 40, 400, 240, 0), (2, 80, 60, 480, 160, 1), (2, 80, 60, 480, 160, 2)])
def test_stft_mel_stays_inside(B, n_mel, n_out, win, step, kind):
    from speech_recognition_amd.features import path_b_tables
    lib = _lib.load()
    t = path_b_tables(win, n_mel, n_out)
    plan = ctypes.c_void_p()
    _lib.check(lib.kws_stft_plan_create(win, step, 512, n_mel, n_out, t['window'].ctypes.data_as(ctypes.c_void_p),
                                        t['mel'].ctypes.data_as(ctypes.c_void_p), t['dct'].ctypes.data_as(ctypes.c_void_p),
                                        1e-6, 0.0, ctypes.byref(plan)), "plan")
    F = lib.kws_stft_num_frames(plan, 16000)
    width = {0: n_out, 1: 257, 2: n_mel}[kind]
    x = Guarded((B, 16000))                  # the INPUT sits between guards too: reads past a clip show up as NaNs
    x.view.copy_(torch.randn(B, 16000, device="cuda") * 0.1)
    out = Guarded((B, F * width))
    _lib.call("kws_stft_mel_f32", plan, _lib.ptr(x.view), B, 16000, _lib.ptr(out.view), kind, S())
    out.check("kws_stft_mel_f32(kind=%d)" % kind)
    assert bool(torch.isfinite(out.view).all())      # nothing from the (NaN-patterned) guards leaked into a frame
    lib.kws_stft_plan_destroy(plan)


@pytest.mark.parametrize("M,K,N", [(1, 128, 128), (129, 128, 128), (1000, 192, 192), (9216, 512, 512), (777, 320, 384)])
def test_gemms_stay_inside(M, K, N):
    lib = _lib.load()
    g = torch.Generator(device="cuda")
    g.manual_seed(M)
    A = torch.randn((M, K), generator=g, device="cuda")
    W = torch.randn((K, N), generator=g, device="cuda") * 0.1
    C = Guarded((M, N))
    stats = Guarded((int(lib.kws_gemm_nn_stats_rows(M, K, N)) * 2 * N,))
    _lib.call("kws_gemm_nn_f32", _lib.ptr(A), _lib.ptr(W), _lib.ptr(C.view), M, K, N, _lib.ptr(stats.view), S())
    C.check("kws_gemm_nn_f32")
    stats.check("kws_gemm_nn_f32 statistics rows")
    G = torch.randn((M, N), generator=g, device="cuda")
    dW = Guarded((K, N))
    ws = Guarded((int(lib.kws_gemm_tn_workspace_floats(M, K, N)),))
    ws.view.zero_()                          # the workspace is scratch: only its margins are checked
    _lib.call("kws_gemm_tn_f32", _lib.ptr(A), _lib.ptr(G), _lib.ptr(dW.view), M, K, N, _lib.ptr(ws.view), S())
    dW.check("kws_gemm_tn_f32")
    torch.cuda.synchronize()
    assert bool((ws.buf[:GUARD] == SENT).all()) and bool((ws.buf[-GUARD:] == SENT).all()), "kws_gemm_tn_f32 left its workspace"


@pytest.mark.parametrize("B,Lin,C,stride", [(3, 397, 128, 1), (2, 22, 384, 2), (5, 11, 512, 1), (1, 199, 192, 2)])
def test_depthwise_forward_stays_inside(B, Lin, C, stride):
    if stride == 1:
        Lout, pad_l = Lin - 2, 0
    else:
        Lout = (Lin + 1) // 2
        pad_l = max((Lout - 1) * 2 + 3 - Lin, 0) // 2
    g = torch.Generator(device="cuda")
    g.manual_seed(Lin)
    y = torch.randn((B, Lin, C), generator=g, device="cuda")
    bn = torch.cat([torch.ones(C, device="cuda"), torch.zeros(3 * C, device="cuda")])
    w = torch.randn((3, C), generator=g, device="cuda")
    z = Guarded((B, Lout, C))
    _lib.call("kws_dwconv_fwd_f32", _lib.ptr(y), _lib.ptr(bn), _lib.ptr(w), _lib.ptr(z.view), B, Lin, Lout, C, stride, pad_l, S())
    z.check("kws_dwconv_fwd_f32")


def test_tail_ops_stay_inside():
    B, T, C, NC = 7, 9, 512, 12
    g = torch.Generator(device="cuda")
    g.manual_seed(1)
    x = torch.rand((B, T, C), generator=g, device="cuda") * 6
    att = torch.softmax(torch.randn((B, T), generator=g, device="cuda"), 1)
    feat = Guarded((B, 2 * C))
    _lib.call("kws_attn_pool_fwd", _lib.ptr(x), _lib.ptr(att), _lib.ptr(feat.view), B, T, C, S())
    feat.check("kws_attn_pool_fwd")
    dr = Guarded((B, 2 * C))
    _lib.call("kws_dropout_fwd", _lib.ptr(feat.view), _lib.ptr(dr.view), B, 2 * C, 0.6, ctypes.c_uint64(3), 0, 2, 0, S())
    dr.check("kws_dropout_fwd")
    p = torch.softmax(torch.randn((B, NC), generator=g, device="cuda"), 1)
    y = torch.eye(NC, device="cuda")[torch.randint(0, NC, (B,), generator=g, device="cuda")]
    per = Guarded((B,))
    _lib.call("kws_softmax_xent_smooth_fwd", _lib.ptr(p), _lib.ptr(y), _lib.ptr(per.view), None, B, NC, 0.1, S())
    per.check("kws_softmax_xent_smooth_fwd")
    p12 = Guarded((B, 12))
    from speech_recognition_amd.model import head32to12
    p32 = torch.softmax(torch.randn((B, 32), generator=g, device="cuda"), 1)
    head32to12(p32, out=p12.view)
    p12.check("kws_head32to12")
