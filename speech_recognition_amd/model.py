"""Host-side mirror of the reference's model.py for the accelerated models: `speech_model`
dispatch (reference model.py:1729-1781), `prepare_model_settings` (model.py:1785-1829) and the two
symbols checkpoints name as custom objects (`relu6` model.py:30-31, `overlapping_time_slice_stack`
model.py:67-76).  The layer graphs themselves are native network programs in csrc/net.hip."""
from . import _lib
from .keras_api import Model, RMSprop
from .net import DeviceNet

ACCELERATED = ('conv_1d_time_sliced_with_attention', 'conv_1d_log_mfcc', 'conv_1d_spectrogram', 'steffeNet', 'conv_1d_residual', 'conv_1d_mfcc_and_raw')
REFERENCE_MODEL_TYPES = (
    'simple', 'snn', 'conv_1d_time_stacked', 'conv_1d_multi_time_sliced', 'conv_1d_time_sliced',
    'conv_1d_time_sliced_group', 'conv_1d_heavy', 'conv_1d_simple', 'conv_1d_gru', 'conv_2d', 'conv_2d_fast',
    'conv_2d_mobile', 'inception', 'inception_d1', 'conv_1d_learned_spec', 'conv_1d_spec', 'conv_1d_fast',
    'conv_1d_top_down', 'conv_1d_residual', 'xception_with_attention', 'conv_1d_time_sliced_with_attention',
    'conv_1d_log_mfcc', 'conv_1d_spectrogram', 'conv_1d_mfcc_and_raw', 'steffeNet')


def relu6(x):
    """K.relu(x, max_value=6).  On the device ReLU6 is fused into the consumer of every BatchNorm
    (csrc/dwconv.hip, csrc/tail.hip); this callable exists for checkpoint custom_objects."""
    return x.clamp(0, 6) if hasattr(x, 'clamp') else min(max(x, 0), 6)


def overlapping_time_slice_stack(x, ksize, stride, padding='SAME'):
    """extract_image_patches framing.  On the device it is fused into the first convolution's
    gathered A-operand (kws_gemm_gather_f32); calling it on host data is not part of the hot path."""
    raise _lib.KwsError("overlapping_time_slice_stack is fused into kws_gemm_gather_f32 on the device")


def class_map_32_to_12(all_classes=None, wanted_classes=None):
    """int32 map [len(all_classes) + 2] -> slot of the 12-class head, as freeze_graph_32_classes.py:55-69 walks it:
    silence -> 0, the unknown-unknown class and every word outside `wanted_classes` -> 1 (their MAXIMUM becomes the
    unknown probability), wanted words -> 2.. in the order they appear in `all_classes`."""
    import numpy as np
    from .classes import get_classes
    all_classes = get_classes(wanted_only=False) if all_classes is None else list(all_classes)
    wanted_classes = get_classes(wanted_only=True) if wanted_classes is None else list(wanted_classes)
    mp = np.zeros(len(all_classes) + 2, np.int32)
    mp[1], slot = 1, 2
    for i, c in enumerate(all_classes):
        if c in wanted_classes:
            mp[i + 2] = slot
            slot += 1
        else:
            mp[i + 2] = 1
    return mp, slot


_HEAD_MAPS = {}


def head32to12(all_probs, all_classes=None, wanted_classes=None, out=None):
    """The 32 -> 12 class head of the reference's frozen graph (freeze_graph_32_classes.py:55-69, BASELINE config C3)
    on the device: `all_probs` [B, 32] softmax outputs of a 32-class model (CUDA tensor / DeviceArray / array) ->
    [B, 12] = softmax over (silence, max over the unknown words, the wanted words).  One `kws_head32to12` launch."""
    import torch
    from .device_array import as_device_f32
    dev = all_probs.device if isinstance(all_probs, torch.Tensor) and all_probs.is_cuda else \
        torch.device("cuda", torch.cuda.current_device())
    p = as_device_f32(all_probs, dev)
    key = (tuple(all_classes) if all_classes is not None else None,
           tuple(wanted_classes) if wanted_classes is not None else None, str(dev))
    if key not in _HEAD_MAPS:
        mp, n_out = class_map_32_to_12(all_classes, wanted_classes)
        _HEAD_MAPS[key] = (torch.from_numpy(mp).to(dev), n_out)
    dmap, n_out = _HEAD_MAPS[key]
    if p.dim() != 2 or p.shape[1] != dmap.numel():
        raise ValueError("head32to12: expected [B, %d] probabilities, got %s" % (dmap.numel(), tuple(p.shape)))
    B = p.shape[0]
    if out is None:
        out = torch.empty((B, n_out), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        _lib.call("kws_head32to12", _lib.ptr(p), p.shape[1], _lib.ptr(dmap), n_out, _lib.ptr(out), B, _lib.stream_ptr())
    return out


def conv_1d_time_sliced_with_attention_model(input_size=16000, num_classes=11, filter_mult=1):
    """reference model.py:775-838: 12-block depthwise/pointwise 1-D CNN on raw waveform, attention-pooled
    head, RMSprop(1e-3), label-smoothed CE (0.1), categorical accuracy."""
    net = DeviceNet(_lib.KWS_NET_TS_ATTENTION, num_classes, filter_mult=filter_mult, input_size=input_size)
    return Model(net, RMSprop(lr=1e-3), name='conv_1d_time_sliced_with_attention')


def conv_1d_log_mfcc_model(input_size=16000, num_classes=11, *args, **kwargs):
    """reference model.py:1400-1479: residual depthwise 1-D CNN on [spectrogram_length, num_log_mel_features]
    features, softmax-over-time attention, RMSprop(6e-4), categorical CE."""
    time_size = kwargs.get('spectrogram_length', 65)
    frequency_size = kwargs.get('num_log_mel_features', 40)
    net = DeviceNet(_lib.KWS_NET_LOG_MFCC, num_classes, input_size=input_size, spectrogram_length=time_size,
                    num_features=frequency_size)
    return Model(net, RMSprop(lr=6e-4), name='conv_1d_log_mfcc', loss='cce')


def conv_1d_spectrogram_model(input_size=16000, num_classes=11, *args, **kwargs):
    """reference model.py:1482-1561: the conv_1d_log_mfcc architecture on the generator's 'spec' output
    ([spectrogram_length, spectrogram_frequencies = 257] magnitudes), RMSprop(3e-4), categorical CE."""
    time_size = kwargs.get('spectrogram_length', 65)
    frequency_size = kwargs.get('spectrogram_frequencies', 257)
    net = DeviceNet(_lib.KWS_NET_LOG_MFCC, num_classes, input_size=input_size, spectrogram_length=time_size,
                    num_features=frequency_size)
    return Model(net, RMSprop(lr=3e-4), name='conv_1d_spectrogram', loss='cce')


def steffeNet(input_size=16000, num_classes=11, *args, **kwargs):
    """reference model.py:1663-1726: raw waveform -> Conv1D(256, 75, strides=50) -> context block -> 12 residual
    depthwise blocks (320 ... 1536 wide, the first depthwise of every other block strided) -> global max ++
    average pooling -> Dropout(.5) -> Dense, RMSprop(1e-3), label-smoothed CE (0.1)."""
    net = DeviceNet(_lib.KWS_NET_STEFFE, num_classes, input_size=input_size)
    return Model(net, RMSprop(lr=1e-3), name='steffeNet')


def conv_1d_residual_model(input_size=16000, num_classes=11, filter_mult=1):
    """reference model.py:841-908: raw waveform -> time-slice stack -> Conv1D(64, 3, strides=2) -> 13 residual blocks
    with 3-wide max-pool joins -> reduce block (1024) -> global average pooling -> Dropout(.5) -> Dense,
    RMSprop(1e-4), categorical CE."""
    net = DeviceNet(_lib.KWS_NET_RESIDUAL, num_classes, filter_mult=filter_mult, input_size=input_size)
    return Model(net, RMSprop(lr=1e-4), name='conv_1d_residual', loss='cce')


def conv_1d_mfcc_and_raw_model(input_size=16000, num_classes=11, *args, **kwargs):
    """reference model.py:1563-1660: the two-input model fed by the generator's 'mfcc_and_raw' output - log-mel
    features -> Conv1D(64, 3) and raw frames (480 / 160, VALID) -> Conv1D(96, 3), concatenated, 10 residual blocks
    with 3-wide max-pool joins, global average pooling, Dropout(.3), Dense; RMSprop(5e-4), categorical CE.
    `input_size` is the feature input's size (as the reference passes it); batches are `[mfcc, raw]`."""
    time_size = kwargs.get('spectrogram_length', 65)
    frequency_size = kwargs.get('num_log_mel_features', 40)
    raw_size = kwargs.get('desired_samples', 16000)
    if kwargs.get('window_size_samples', 480) != 480 or kwargs.get('window_stride_samples', 160) != 160:
        raise NotImplementedError("conv_1d_mfcc_and_raw: only the 30 ms / 10 ms framing (480 / 160 samples) is built")
    net = DeviceNet(_lib.KWS_NET_MFCC_AND_RAW, num_classes, input_size=time_size * frequency_size + raw_size,
                    spectrogram_length=time_size, num_features=frequency_size)
    return Model(net, RMSprop(lr=5e-4), name='conv_1d_mfcc_and_raw', loss='cce')


def speech_model(model_type, input_size, num_classes=11, *args, **kwargs):
    if model_type == 'conv_1d_time_sliced_with_attention':
        return conv_1d_time_sliced_with_attention_model(input_size, num_classes)
    if model_type == 'conv_1d_log_mfcc':
        return conv_1d_log_mfcc_model(input_size, num_classes, *args, **kwargs)
    if model_type == 'conv_1d_spectrogram':
        return conv_1d_spectrogram_model(input_size, num_classes, *args, **kwargs)
    if model_type == 'steffeNet':
        return steffeNet(input_size, num_classes, *args, **kwargs)
    if model_type == 'conv_1d_residual':
        return conv_1d_residual_model(input_size, num_classes)
    if model_type == 'conv_1d_mfcc_and_raw':
        return conv_1d_mfcc_and_raw_model(input_size, num_classes, *args, **kwargs)
    if model_type in REFERENCE_MODEL_TYPES:
        raise NotImplementedError(
            "model '%s' is outside the accelerated hot path (SURVEY.md 8: only %s are built natively)"
            % (model_type, ', '.join(ACCELERATED)))
    raise ValueError("Invalid model: %s" % model_type)


def prepare_model_settings(label_count, sample_rate, clip_duration_ms, window_size_ms, window_stride_ms,
                           dct_coefficient_count, num_log_mel_features, output_representation='raw'):
    """Settings arithmetic of reference model.py:1785-1829 (truncating int() conversions included)."""
    desired_samples = int(sample_rate * clip_duration_ms / 1000)
    window_size_samples = int(sample_rate * window_size_ms / 1000)
    window_stride_samples = int(sample_rate * window_stride_ms / 1000)
    length_minus_window = desired_samples - window_size_samples
    spectrogram_frequencies = 257
    spectrogram_length = 0 if length_minus_window < 0 else 1 + int(length_minus_window / window_stride_samples)
    fingerprint_size = {
        'mfcc': num_log_mel_features * spectrogram_length,
        'raw': desired_samples,
        'spec': spectrogram_frequencies * spectrogram_length,
        'mfcc_and_raw': num_log_mel_features * spectrogram_length,
    }[output_representation]
    return {
        'desired_samples': desired_samples,
        'window_size_samples': window_size_samples,
        'window_stride_samples': window_stride_samples,
        'spectrogram_length': spectrogram_length,
        'spectrogram_frequencies': spectrogram_frequencies,
        'dct_coefficient_count': dct_coefficient_count,
        'fingerprint_size': fingerprint_size,
        'label_count': label_count,
        'sample_rate': sample_rate,
        'num_log_mel_features': num_log_mel_features,
    }
