"""Host-side owner of one network program of libkws_hip.so.

Holds the flat parameter / state / gradient / optimizer-slot buffers and the workspace as
PyTorch-ROCm tensors (device memory + streams only) and drives ``kws_net_*``.  The layer math
itself is in csrc/; this file is plumbing.
"""
import ctypes
from collections import OrderedDict

import numpy as np
import torch

from . import _lib
from ._lib import KWS_NET_LOG_MFCC, KWS_NET_TS_ATTENTION  # noqa: F401


class TensorSpec(object):
    __slots__ = ("name", "offset", "size", "shape", "is_state", "l2", "fan_in", "fan_out", "init")

    def __repr__(self):
        return "TensorSpec(%s %s @%d)" % (self.name, self.shape, self.offset)


class DeviceNet(object):
    """One model replica on one GPU."""

    def __init__(self, kind, num_classes, filter_mult=1, input_size=16000, spectrogram_length=0,
                 num_features=0, device=None, seed=87654321, gemm_mode=None):
        if not torch.cuda.is_available():
            raise _lib.KwsError("no MI355X visible to this process: the HIP path is the only path "
                                "(no CPU fallback)")
        self.lib = _lib.load()
        self.device = torch.device(device if device is not None else "cuda:%d" % torch.cuda.current_device())
        cfg = _lib.NetConfig(kind, num_classes, filter_mult, input_size, spectrogram_length, num_features)
        h = ctypes.c_void_p()
        _lib.check(self.lib.kws_net_create(ctypes.byref(cfg), ctypes.byref(h)), "kws_net_create")
        self.handle = h
        self.kind = kind
        self.num_classes = num_classes
        self.input_size = input_size
        self.spectrogram_length, self.num_features = spectrogram_length, num_features
        self.n_params = int(self.lib.kws_net_num_params(h))
        self.n_state = int(self.lib.kws_net_num_state(h))
        self.tensors = OrderedDict()
        for i in range(self.lib.kws_net_num_tensors(h)):
            ti = _lib.TensorInfo()
            _lib.check(self.lib.kws_net_tensor_info(h, i, ctypes.byref(ti)), "kws_net_tensor_info")
            s = TensorSpec()
            s.name = ti.name.decode()
            s.offset, s.size = int(ti.offset), int(ti.size)
            s.shape = tuple(int(ti.shape[k]) for k in range(ti.ndim))
            s.is_state = bool(ti.is_state)
            s.l2, s.fan_in, s.fan_out, s.init = float(ti.l2), ti.fan_in, ti.fan_out, float(ti.init)
            self.tensors[s.name] = s
        with torch.cuda.device(self.device):
            self.params = torch.zeros(self.n_params, dtype=torch.float32, device=self.device)
            self.state = torch.zeros(self.n_state, dtype=torch.float32, device=self.device)
            self.grads = torch.zeros(self.n_params, dtype=torch.float32, device=self.device)
            self.slots = torch.zeros(self.n_params, dtype=torch.float32, device=self.device)
            self.l2 = torch.zeros(self.n_params, dtype=torch.float32, device=self.device)
            self.metrics = torch.zeros(4, dtype=torch.float32, device=self.device)
            self.reg_loss = torch.zeros(1, dtype=torch.float32, device=self.device)
        self._ws = None
        self._ws_key = None
        # arithmetic of the pointwise GEMMs, a property of THIS handle: 0 = f32 MFMA (the product path), 2 = the fp16 x 2
        # A/B arm.  KWS_GEMM_F16X2=1 only picks the default of nets that do not say (test / profiling runs of the arm).
        if gemm_mode is None:
            import os
            gemm_mode = 2 if os.environ.get("KWS_GEMM_F16X2") else 0
        self.set_gemm_mode(gemm_mode)
        self.initialize(seed)

    @property
    def gemm_mode(self):
        return int(self.lib.kws_net_get_gemm_mode(self.handle))

    def set_gemm_mode(self, mode):
        _lib.check(self.lib.kws_net_set_gemm_mode(self.handle, int(mode)), "kws_net_set_gemm_mode")

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                self.lib.kws_net_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

    # -- parameters ------------------------------------------------------------------------------
    def count_params(self):
        return sum(s.size for s in self.tensors.values())

    def trainable_count(self):
        return sum(s.size for s in self.tensors.values() if not s.is_state)

    def initialize(self, seed=87654321):
        """Keras initialisers: glorot_uniform kernels, ones/zeros for BN, zeros bias."""
        rng = np.random.RandomState(seed)
        p = np.zeros(self.n_params, np.float32)
        st = np.zeros(self.n_state, np.float32)
        l2 = np.zeros(self.n_params, np.float32)
        for s in self.tensors.values():
            dst = st if s.is_state else p
            if s.fan_in > 0:
                limit = np.sqrt(6.0 / (s.fan_in + s.fan_out))
                dst[s.offset:s.offset + s.size] = rng.uniform(-limit, limit, size=s.size).astype(np.float32)
            else:
                dst[s.offset:s.offset + s.size] = s.init
            if not s.is_state:
                l2[s.offset:s.offset + s.size] = s.l2
        self.params.copy_(torch.from_numpy(p))
        self.state.copy_(torch.from_numpy(st))
        self.l2.copy_(torch.from_numpy(l2))
        self.slots.zero_()
        self.grads.zero_()

    def get_weights(self):
        """OrderedDict name -> numpy array (Keras variable names and shapes)."""
        p = self.params.cpu().numpy()
        st = self.state.cpu().numpy()
        out = OrderedDict()
        for s in self.tensors.values():
            src = st if s.is_state else p
            out[s.name] = src[s.offset:s.offset + s.size].reshape(s.shape).copy()
        return out

    def set_weights(self, weights):
        p = self.params.cpu().numpy()
        st = self.state.cpu().numpy()
        for name, arr in weights.items():
            s = self.tensors[name]
            arr = np.asarray(arr, dtype=np.float32)
            if arr.size != s.size:
                raise ValueError("%s: expected %s, got %s" % (name, s.shape, arr.shape))
            (st if s.is_state else p)[s.offset:s.offset + s.size] = arr.reshape(-1)
        self.params.copy_(torch.from_numpy(p))
        self.state.copy_(torch.from_numpy(st))

    def grads_dict(self):
        g = self.grads.cpu().numpy()
        return OrderedDict((s.name, g[s.offset:s.offset + s.size].reshape(s.shape).copy())
                           for s in self.tensors.values() if not s.is_state)

    # -- execution ---------------------------------------------------------------------------------
    def _workspace(self, B, training):
        key = (int(training),)
        need = int(self.lib.kws_net_workspace_bytes(self.handle, B, int(training)))
        if self._ws is None or self._ws.numel() * 4 < need or self._ws_key != key:
            self._ws = None
            self._ws = torch.empty((need + 3) // 4, dtype=torch.float32, device=self.device)
            self._ws_key = key
        return self._ws

    def predict(self, x, out=None, stream=None):
        """x: float32 CUDA tensor [B, input_size] -> probabilities [B, num_classes]."""
        B = x.shape[0]
        ws = self._workspace(B, False)
        if out is None:
            out = torch.empty((B, self.num_classes), dtype=torch.float32, device=self.device)
        _lib.call("kws_net_predict", self.handle, _lib.ptr(self.params), _lib.ptr(self.state), _lib.ptr(x),
                  B, _lib.ptr(out), _lib.ptr(ws), ws.numel() * 4, _lib.stream_ptr(stream))
        return out

    def train_fwd_bwd(self, x, y, seed, step, row_offset=0, loss_batch=None, probs=None, stream=None):
        """Forward + backward of one batch; fills self.grads / self.metrics, updates BN moving stats."""
        B = x.shape[0]
        ws = self._workspace(B, True)
        if probs is None:
            probs = torch.empty((B, self.num_classes), dtype=torch.float32, device=self.device)
        _lib.call("kws_net_train_fwd_bwd", self.handle, _lib.ptr(self.params), _lib.ptr(self.state),
                  _lib.ptr(x), _lib.ptr(y), B, _lib.ptr(self.grads), _lib.ptr(probs), _lib.ptr(self.metrics),
                  ctypes.c_uint64(seed), ctypes.c_uint32(step), row_offset,
                  B if loss_batch is None else loss_batch, _lib.ptr(ws), ws.numel() * 4,
                  _lib.stream_ptr(stream))
        return probs

    def train_fwd_bwd_part(self, part, split_block, x, y, seed, step, row_offset=0, loss_batch=None, probs=None, stream=None):
        """Half of train_fwd_bwd (kws_net_train_fwd_bwd_part): part 1 = forward + backward down to `split_block`, part 2 =
        the rest.  After part 1, self.grads[self.grad_ready_offset(split_block):] is final."""
        B = x.shape[0]
        ws = self._workspace(B, True)
        if probs is None:
            probs = self._probs_part if part == 2 and getattr(self, "_probs_part", None) is not None else \
                torch.empty((B, self.num_classes), dtype=torch.float32, device=self.device)
        self._probs_part = probs if part == 1 else None
        _lib.call("kws_net_train_fwd_bwd_part", self.handle, _lib.ptr(self.params), _lib.ptr(self.state), _lib.ptr(x),
                  _lib.ptr(y), B, _lib.ptr(self.grads), _lib.ptr(probs), _lib.ptr(self.metrics), ctypes.c_uint64(seed),
                  ctypes.c_uint32(step), row_offset, B if loss_batch is None else loss_batch, _lib.ptr(ws), ws.numel() * 4,
                  int(part), int(split_block), _lib.stream_ptr(stream))
        return probs

    def num_blocks(self):
        return int(self.lib.kws_net_num_blocks(self.handle))

    def grad_ready_offset(self, split_block):
        off = int(self.lib.kws_net_grad_ready_offset(self.handle, int(split_block)))
        if off < 0:
            raise _lib.KwsError("grad_ready_offset: split_block %d is not valid for this network" % split_block)
        return off

    def debug_view(self, B, what, index, training=True):
        """Copy of one intermediate tensor of the LAST call with this (B, training) (parity tests)."""
        off, cnt = ctypes.c_int64(), ctypes.c_int64()
        _lib.call("kws_net_debug_view", self.handle, B, int(training), what, index, ctypes.byref(off),
                  ctypes.byref(cnt))
        return self._ws[off.value:off.value + cnt.value].cpu().numpy()

    def l2_loss(self, stream=None):
        _lib.call("kws_l2_loss", _lib.ptr(self.params), _lib.ptr(self.l2), self.n_params,
                  _lib.ptr(self.reg_loss), _lib.stream_ptr(stream))
        return self.reg_loss

    def rmsprop_step(self, lr, rho=0.9, eps=1e-8, grad_scale=1.0, stream=None):
        _lib.call("kws_rmsprop_step", _lib.ptr(self.params), _lib.ptr(self.grads), _lib.ptr(self.slots),
                  _lib.ptr(self.l2), self.n_params, lr, rho, eps, grad_scale, _lib.stream_ptr(stream))

    def sgd_step(self, lr, momentum=0.9, grad_scale=1.0, stream=None):
        _lib.call("kws_sgd_momentum_step", _lib.ptr(self.params), _lib.ptr(self.grads), _lib.ptr(self.slots),
                  _lib.ptr(self.l2), self.n_params, lr, momentum, grad_scale, _lib.stream_ptr(stream))
