"""ctypes binding of libkws_hip.so (the C ABI declared in include/kws_hip.h).

The product path has NO CPU fallback: if the shared library is missing or a call fails, a
KwsError is raised.  PyTorch-ROCm is used only to own device memory and streams; every
pointer handed to the library is a tensor's ``data_ptr()``.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("KWS_LIB_PATH") or os.path.join(_HERE, "libkws_hip.so")  # override: kernel A/B experiments

ABI_VERSION = 5    # include/kws_hip.h: KWS_ABI_VERSION

KWS_NET_TS_ATTENTION = 1
KWS_NET_LOG_MFCC = 2
KWS_NET_STEFFE = 3
KWS_NET_RESIDUAL = 4
KWS_NET_MFCC_AND_RAW = 5


class KwsError(RuntimeError):
    pass


class GatherDesc(ctypes.Structure):
    _fields_ = [("L_out", ctypes.c_int), ("cin", ctypes.c_int), ("taps", ctypes.c_int),
                ("stride_t", ctypes.c_int), ("stride_j", ctypes.c_int), ("base_off", ctypes.c_int),
                ("x_len", ctypes.c_int), ("x_batch_stride", ctypes.c_int64)]


class SamplerSet(ctypes.Structure):
    _fields_ = [("rows", ctypes.c_void_p), ("labels", ctypes.c_void_p), ("silence", ctypes.c_void_p),
                ("n", ctypes.c_int32)]


class SamplerArgs(ctypes.Structure):
    _fields_ = [("deterministic", ctypes.c_int32), ("offset", ctypes.c_int32), ("count", ctypes.c_int32),
                ("use_background", ctypes.c_int32), ("n_bg", ctypes.c_int32), ("bg_len", ctypes.c_void_p),
                ("bg_start", ctypes.c_void_p), ("desired_samples", ctypes.c_int32), ("shift_lo", ctypes.c_int32),
                ("shift_hi", ctypes.c_int32), ("background_frequency", ctypes.c_double),
                ("background_volume_range", ctypes.c_double), ("foreground_frequency", ctypes.c_double),
                ("foreground_volume_range", ctypes.c_double), ("time_shift_frequency", ctypes.c_double),
                ("pseudo_frequency", ctypes.c_double), ("flip_frequency", ctypes.c_double),
                ("silence_volume_range", ctypes.c_double)]


class NetConfig(ctypes.Structure):
    _fields_ = [("kind", ctypes.c_int), ("num_classes", ctypes.c_int), ("filter_mult", ctypes.c_int),
                ("input_size", ctypes.c_int), ("spectrogram_length", ctypes.c_int),
                ("num_features", ctypes.c_int)]


class TensorInfo(ctypes.Structure):
    _fields_ = [("name", ctypes.c_char * 64), ("offset", ctypes.c_int64), ("size", ctypes.c_int64),
                ("ndim", ctypes.c_int), ("shape", ctypes.c_int64 * 4), ("is_state", ctypes.c_int),
                ("l2", ctypes.c_float), ("fan_in", ctypes.c_int), ("fan_out", ctypes.c_int),
                ("init", ctypes.c_float)]


_P = ctypes.c_void_p
_I = ctypes.c_int
_I64 = ctypes.c_int64
_F = ctypes.c_float

# name -> (restype, argtypes); every symbol include/kws_hip.h declares
SIGNATURES = {
    "kws_abi_version": (_I, []),
    "kws_last_error": (ctypes.c_char_p, []),
    "kws_device_name": (_I, [ctypes.c_char_p, _I]),
    "kws_stream_create": (_I, [_I, ctypes.POINTER(_P)]),
    "kws_stream_destroy": (_I, [_P]),
    "kws_profiler_create": (_I, [ctypes.POINTER(_P)]),
    "kws_profiler_destroy": (_I, [_P]),
    "kws_profiler_attach": (_I, [_P]),
    "kws_profiler_collect": (_I, [_P]),
    "kws_profiler_get": (_I, [_P, _I, ctypes.c_char_p, _I, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(_I64),
                              ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]),
    "kws_sampler_draw": (_I, [_P, ctypes.POINTER(_I), ctypes.POINTER(SamplerSet), ctypes.POINTER(SamplerSet),
                              ctypes.POINTER(SamplerArgs), _P, _P, _P, _P, _P, _P]),
    "kws_augment_f32": (_I, [_P, _I64, _I, _P, _P, _P, _P, _I64, _P, _P, _P, _I, _P]),
    "kws_augment_i16": (_I, [_P, _I64, _I, _P, _P, _P, _P, _I64, _P, _P, _P, _I, _P]),
    "kws_tta_transform": (_I, [_P, _P, _I, _I, _I, _P]),
    "kws_tta_combine": (_I, [ctypes.POINTER(_P), _I, _F, _P, _P, _I, _I, _P]),
    "kws_head32to12": (_I, [_P, _I, _P, _I, _P, _I, _P]),
    "kws_dropout_fwd": (_I, [_P, _P, _I, _I, _F, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint32, _I64, _P]),
    "kws_dropout_bwd": (_I, [_P, _P, _I, _I, _F, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint32, _I64, _P]),
    "kws_attn_pool_fwd": (_I, [_P, _P, _P, _I, _I, _I, _P]),
    "kws_attn_pool_bwd_workspace_floats": (_I64, [_I, _I, _I]),
    "kws_attn_pool_bwd": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "kws_softmax_xent_smooth_fwd": (_I, [_P, _P, _P, _P, _I, _I, _F, _P]),
    "kws_softmax_xent_smooth_bwd": (_I, [_P, _P, _P, _P, _I, _I, _F, _F, _P]),
    "kws_comm_unique_id": (_I, [_P]),
    "kws_comm_create": (_I, [_I, _I, _P, ctypes.POINTER(_P)]),
    "kws_comm_destroy": (_I, [_P]),
    "kws_allreduce_grads": (_I, [_P, _P, _I64, _P]),
    "kws_stretch_plan_create": (_I, [_I, ctypes.c_double, ctypes.POINTER(_P)]),
    "kws_stretch_plan_destroy": (_I, [_P]),
    "kws_stretch_out_samples": (_I, [_P]),
    "kws_time_stretch_f32": (_I, [_P, _P, _F, _P, _I, _I, _I, _P]),
    "kws_time_stretch_i16": (_I, [_P, _P, _P, _I, _I, _I, _P]),
    "kws_stft_plan_create": (_I, [_I, _I, _I, _I, _I, _P, _P, _P, _F, _F, ctypes.POINTER(_P)]),
    "kws_stft_plan_destroy": (_I, [_P]),
    "kws_stft_num_frames": (_I, [_P, _I]),
    "kws_stft_mel_f32": (_I, [_P, _P, _I, _I, _P, _I, _P]),
    "kws_gemm_num_row_tiles": (_I, [_I64]),
    "kws_gemm_nn_stats_rows": (_I, [_I64, _I, _I]),
    "kws_gemm_gather_stats_rows": (_I, [_I64]),
    "kws_gemm_nn_f32": (_I, [_P, _P, _P, _I64, _I, _I, _P, _P]),
    "kws_net_get_gemm_mode": (_I, [_P]),
    "kws_net_set_gemm_mode": (_I, [_P, _I]),
    "kws_gemm_nn_f16x2_supported": (_I, [_I64, _I, _I]),
    "kws_gemm_tn_f16x2_supported": (_I, [_I64, _I, _I]),
    "kws_gemm_nn_f16x2_stats_rows": (_I, [_I64]),
    "kws_absmax_batch_f32": (_I, [_P, _P, _P, _I, _P]),
    "kws_dwconv_fwd_amax_f32": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _P]),
    "kws_dwconv_bwd_bn_amax_f32": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _P]),
    "kws_bn_bwd_apply_amax": (_I, [_P, _P, _P, _P, _P, _I64, _I, _P, _P]),
    "kws_f16x2_split_batch": (_I, [_P, _P, _P, _P, _P, _P, _I, _P]),
    "kws_gemm_nn_f16x2_f32": (_I, [_P, _P, _P, _I64, _I, _I, _P, _P, _P, _P]),
    "kws_gemm_tn_f16x2_workspace_floats": (_I64, [_I64, _I, _I]),
    "kws_gemm_tn_f16x2_f32": (_I, [_P, _P, _P, _I64, _I, _I, _P, _P, _P, _P]),
    "kws_gemm_gather_f32": (_I, [_P, ctypes.POINTER(GatherDesc), _P, _P, _I, _I, _P, _P]),
    "kws_gemm_tn_workspace_floats": (_I64, [_I64, _I, _I]),
    "kws_gemm_tn_f32": (_I, [_P, _P, _P, _I64, _I, _I, _P, _P]),
    "kws_gemm_tn_gather_f32": (_I, [_P, ctypes.POINTER(GatherDesc), _P, _P, _I, _I, _P, _P]),
    "kws_transpose_f32": (_I, [_P, _P, _I, _I, _P]),
    "kws_bn_stats_finalize": (_I, [_P, _I, _I64, _I, _P, _P, _F, _F, _P, _P, _P, _P, _P]),
    "kws_bn_infer_prepare": (_I, [_P, _P, _P, _P, _F, _I, _P, _P]),
    "kws_bn_relu6_apply": (_I, [_P, _P, _P, _I64, _I, _I, _P]),
    "kws_dwconv_fwd_f32": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "kws_dwconv_bwd_part_floats": (_I64, [_I, _I, _I]),
    "kws_dwconv_bwd_f32": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "kws_dwconv_bwd_bn_f32": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "kws_dw_bwd_finalize": (_I, [_P, _I, _I64, _I, _P, _P, _P, _P, _P, _P]),
    "kws_bn_bwd_apply": (_I, [_P, _P, _P, _P, _P, _I64, _I, _P]),
    "kws_rmsprop_step": (_I, [_P, _P, _P, _P, _I64, _F, _F, _F, _F, _P]),
    "kws_sgd_momentum_step": (_I, [_P, _P, _P, _P, _I64, _F, _F, _F, _P]),
    "kws_l2_loss": (_I, [_P, _P, _I64, _P, _P]),
    "kws_net_create": (_I, [ctypes.POINTER(NetConfig), ctypes.POINTER(_P)]),
    "kws_net_destroy": (_I, [_P]),
    "kws_net_num_params": (_I64, [_P]),
    "kws_net_num_state": (_I64, [_P]),
    "kws_net_num_tensors": (_I, [_P]),
    "kws_net_tensor_info": (_I, [_P, _I, ctypes.POINTER(TensorInfo)]),
    "kws_net_workspace_bytes": (_I64, [_P, _I, _I]),
    "kws_net_debug_view": (_I, [_P, _I, _I, _I, _I, ctypes.POINTER(_I64), ctypes.POINTER(_I64)]),
    "kws_net_predict": (_I, [_P, _P, _P, _P, _I, _P, _P, _I64, _P]),
    "kws_net_train_fwd_bwd": (_I, [_P, _P, _P, _P, _P, _I, _P, _P, _P, ctypes.c_uint64, ctypes.c_uint32,
                                   _I64, _I, _P, _I64, _P]),
    "kws_net_num_blocks": (_I, [_P]),
    "kws_net_grad_ready_offset": (_I64, [_P, _I]),
    "kws_net_train_fwd_bwd_part": (_I, [_P, _P, _P, _P, _P, _I, _P, _P, _P, ctypes.c_uint64, ctypes.c_uint32,
                                        _I64, _I, _P, _I64, _I, _I, _P]),
}

_lib = None


def load():
    """Load libkws_hip.so (built in-tree by __graft_entry__.build() / csrc/Makefile)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise KwsError("libkws_hip.so not found at %s - build it with "
                       "`python -c 'import __graft_entry__ as g; g.build()'` "
                       "(there is no CPU fallback)" % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    lib.kws_abi_version.restype = _I
    if lib.kws_abi_version() != ABI_VERSION:      # before any other symbol is bound: a stale library says so itself
        raise KwsError("%s has ABI version %d, this package expects %d - rebuild it (__graft_entry__.build())" %
                       (LIB_PATH, lib.kws_abi_version(), ABI_VERSION))
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here = header/library mismatch
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        msg = load().kws_last_error()
        raise KwsError("%s failed (%d): %s" % (what or "libkws_hip call", rc,
                                               msg.decode() if msg else "?"))


def ptr(t):
    """Device pointer of a torch tensor (or None)."""
    if t is None:
        return None
    return ctypes.c_void_p(t.data_ptr())


def stream_ptr(stream=None):
    import torch
    s = stream if stream is not None else torch.cuda.current_stream()
    return ctypes.c_void_p(s.cuda_stream)


class OwnedStream(object):
    """A HIP stream of scheduling class `cls` (-1 low, 0 normal, +1 high) created by kws_stream_create, viewed by torch as
    an ExternalStream (which does not own the handle) and destroyed by close() / the finaliser after a synchronise."""

    def __init__(self, device, cls):
        import torch
        self.device = device
        with torch.cuda.device(device):
            h = ctypes.c_void_p()
            check(load().kws_stream_create(int(cls), ctypes.byref(h)), "kws_stream_create")
            self.handle = h.value
            self.stream = torch.cuda.ExternalStream(h.value, device=device)

    def close(self):
        h, self.handle = self.handle, None
        if h and _lib is not None:
            try:
                import torch
                with torch.cuda.device(self.device):
                    self.stream.synchronize()
                    _lib.kws_stream_destroy(ctypes.c_void_p(h))
            except Exception:      # interpreter shutdown: the runtime may be gone already
                pass
        self.stream = None

    def __del__(self):
        self.close()


class Profiler(object):
    """Per-kernel-family HIP-event profiler (kws_profiler_*): a handle that threads attach to.  attach() makes the CALLING
    thread's launches record into it, detach() stops that; collect() -> {family: {"ms", "count", "flops", "bytes"}} of
    everything recorded since the last collect()."""

    def __init__(self):
        h = ctypes.c_void_p()
        check(load().kws_profiler_create(ctypes.byref(h)), "kws_profiler_create")
        self.handle = h

    def attach(self):
        check(load().kws_profiler_attach(self.handle), "kws_profiler_attach")

    @staticmethod
    def detach():
        check(load().kws_profiler_attach(None), "kws_profiler_attach")

    def collect(self):
        lib = load()
        out = {}
        n = lib.kws_profiler_collect(self.handle)
        for i in range(n):
            name = ctypes.create_string_buffer(64)
            ms, fl, by = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
            cnt = ctypes.c_int64()
            check(lib.kws_profiler_get(self.handle, i, name, 64, ctypes.byref(ms), ctypes.byref(cnt), ctypes.byref(fl),
                                       ctypes.byref(by)), "kws_profiler_get")
            out[name.value.decode()] = {"ms": ms.value, "count": cnt.value, "flops": fl.value, "bytes": by.value}
        return out

    def close(self):
        """Destroys the handle.  Raises (and keeps the handle) while another thread is still attached to it."""
        h = self.handle
        if h is not None and _lib is not None:
            check(_lib.kws_profiler_destroy(h), "kws_profiler_destroy")
        self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def call(name, *args):
    lib = load()
    rc = getattr(lib, name)(*args)
    check(rc, name)
