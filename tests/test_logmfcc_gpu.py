"""GPU parity of the conv_1d_log_mfcc network program (BASELINE config C3: 32-class
head on 98x40 features) against the CPU oracle, plus the composed config-C3 inference path
(features -> net -> 32->12 head).  Same method as tests/test_net_gpu.py: the device's discrete decisions
(ReLU6 masks, max-pool winners) are read back and handed to the oracle's backward pass."""
import ctypes

import numpy as np
import pytest
import torch

from oracle import features as OF
from oracle import layers as OL
from oracle.net import LogMfccNet
from speech_recognition_amd import _lib
from speech_recognition_amd.net import DeviceNet

pytestmark = pytest.mark.gpu


def _pair(num_classes=32, seed=11, F=40, T=98):
    ora = LogMfccNet(num_classes=num_classes, spectrogram_length=T, num_features=F, dtype=np.float64)
    rng = np.random.RandomState(seed)
    for k in ora.params:
        if k.endswith('gamma'):
            ora.params[k] = (1.0 + 0.1 * rng.randn(*ora.params[k].shape)).astype(np.float32)
        if k.endswith('beta') or k.endswith('bias'):
            ora.params[k] = (0.1 * rng.randn(*ora.params[k].shape)).astype(np.float32)
    for k in ora.state:
        if k.endswith('moving_mean'):
            ora.state[k] = (0.05 * rng.randn(*ora.state[k].shape)).astype(np.float32)
        else:
            ora.state[k] = (1.0 + 0.2 * rng.rand(*ora.state[k].shape)).astype(np.float32)
    net = DeviceNet(_lib.KWS_NET_LOG_MFCC, num_classes, input_size=T * F, spectrogram_length=T, num_features=F)
    net.set_weights(dict(ora.params, **ora.state))
    return ora, net


def _batch(B, nc, seed, F=40, T=98):
    rng = np.random.RandomState(seed)
    lab = rng.randint(0, nc, B)
    x = rng.randn(B, T, F) * 2.0 - 0.7 + 0.5 * np.sin(np.arange(F)[None, None, :] * (1 + lab)[:, None, None] * 0.1)
    return x.reshape(B, -1).astype(np.float32), np.eye(nc, dtype=np.float32)[lab]


def _decisions(net, ora, B):
    """ReLU6 masks per BN index and max-pool winners per strided block from the device tensors (f32 math
    of the kernels: pre = fmaf(y, scale, shift))."""
    shapes = {ora.first[1]: (B, ora.T0 - 2, 64), ora.att[2]: (B, ora.T, 1)}
    pools = {}
    for i, blk in enumerate(ora.blocks):
        if 'short' in blk:
            shapes[blk['short'][1]] = None                       # linear BN: no mask
        shapes[blk['bn1']] = (B, blk['Lin'], blk['nf'])
        shapes[blk['bn2']] = (B, blk['Lin'], blk['nf'])
        if blk['stride'] != 1:
            pools[i] = blk['bn2']
    masks, pre_of = {}, {}
    for idx, shp in shapes.items():
        if shp is None:
            continue
        C = shp[2]
        bn = net.debug_view(B, 2, idx)
        if idx == ora.att[2]:
            y = net.debug_view(B, 4, 0).reshape(shp)
        else:
            y = net.debug_view(B, 0, idx).reshape(shp)
        pre = (y.astype(np.float64) * bn[:C].astype(np.float64) + bn[C:2 * C].astype(np.float64)).astype(np.float32)
        masks[idx] = ((pre > 0) & (pre <= 6)).astype(np.float64)
        pre_of[idx] = pre
    args = {}
    for i, idx in pools.items():
        a = np.minimum(np.maximum(pre_of[idx], np.float32(0)), np.float32(6))
        Bc, L, C = a.shape
        if L % 2:                                                        # 'same' pooling: the short last window's only
            a = np.concatenate([a, np.full((Bc, 1, C), -np.inf, np.float32)], axis=1)   # element wins
        w = a.reshape(Bc, (L + 1) // 2, 2, C)
        args[i] = (w[:, :, 1, :] > w[:, :, 0, :]).astype(np.int64)      # first maximum wins
    return masks, args


def test_tensor_table_matches_oracle():
    ora, net = _pair()
    assert net.count_params() == 784484
    assert [s.name for s in net.tensors.values() if not s.is_state] == list(ora.params.keys())
    for k, v in list(ora.params.items()) + list(ora.state.items()):
        assert net.tensors[k].shape == v.shape, k


@pytest.mark.parametrize("nc,F", [(32, 40), (12, 257)])
def test_predict_matches_oracle(nc, F):
    ora, net = _pair(nc, F=F)
    x, _ = _batch(7, nc, 1, F)
    p = net.predict(torch.from_numpy(x).cuda()).cpu().numpy()
    ref = ora.forward(x.astype(np.float64), training=False)
    assert np.abs(p - ref).max() < 1e-5
    assert np.array_equal(p.argmax(1), ref.argmax(1))


# F = 257, 12 classes: conv_1d_spectrogram (model.py:1482-1561) - the same program on the
# generator's 'spec' output; 257 is not a multiple of the gathered GEMM's 16-byte vectors, so the first
# convolution runs on re-pitched copies (net_logmfcc.hip: pad_first_conv)
@pytest.mark.parametrize("B,nc,F", [(4, 32, 40), (19, 32, 40), (5, 12, 257)])
def test_train_fwd_bwd_matches_oracle(B, nc, F):
    ora, net = _pair(nc, F=F)
    x, y = _batch(B, nc, B, F)
    probs = net.train_fwd_bwd(torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda(), seed=77, step=2)
    torch.cuda.synchronize()
    masks, args = _decisions(net, ora, B)
    loss, p, grads, cache = ora.loss_and_grads(x.astype(np.float64), y.astype(np.float64), seed=77, step=2,
                                               relu_masks=masks, pool_args=args)
    got = probs.cpu().numpy()
    assert np.abs(got - p).max() < 2e-5
    assert np.array_equal(got.argmax(1), p.argmax(1))
    m = net.metrics.cpu().numpy()
    assert abs(m[0] / B - loss) < 5e-5
    assert m[1] == (p.argmax(1) == y.argmax(1)).sum()
    g = net.grads_dict()
    for k, ref in grads.items():
        if k in ora.l2_names:
            ref = ref - 2e-5 * ora.params[k].astype(np.float64)
        ref = ref.reshape(g[k].shape)
        err = np.abs(g[k] - ref).max() / max(np.abs(ref).max(), 1e-7)
        assert err < 1e-4, (k, err)
    w = net.get_weights()
    for idx, (mean, var) in cache['batch_stats'].items():
        mm = ora.state['batch_normalization_%d/moving_mean' % idx].astype(np.float64)
        np.testing.assert_allclose(w['batch_normalization_%d/moving_mean' % idx], mm - (mm - mean) * 0.01, atol=5e-6)


def test_config_c3_features_net_head32to12():
    """BASELINE config C3 end to end on the device: raw clips -> STFT/mel/DCT (M=40, K=40) -> 32-class
    log-mfcc net -> 32->12 head (freeze_graph_32_classes.py:55-69); checked against the oracle chain."""
    ora, net = _pair()
    rng = np.random.RandomState(3)
    B = 6
    t = np.arange(16000) / 16000.0
    clips = (rng.randn(B, 16000) * 0.0774 + 0.05 * np.sin(2 * np.pi * 300 * t)[None]).astype(np.float32)
    tables = OF.tables_path_b(480, 40, 40)
    lib = _lib.load()
    plan = ctypes.c_void_p()
    win, mel, dct = (np.ascontiguousarray(tables[k], dtype=np.float32) for k in ('window', 'mel', 'dct'))
    _lib.check(lib.kws_stft_plan_create(480, 160, 512, 40, 40, win.ctypes.data_as(ctypes.c_void_p),
                                        mel.ctypes.data_as(ctypes.c_void_p), dct.ctypes.data_as(ctypes.c_void_p),
                                        1e-6, 0.0, ctypes.byref(plan)), "plan")
    dclips = torch.from_numpy(clips).cuda()
    feats = torch.empty((B, 98 * 40), device="cuda")
    _lib.call("kws_stft_mel_f32", plan, _lib.ptr(dclips), B, 16000, _lib.ptr(feats), 0, _lib.stream_ptr())
    p32 = net.predict(feats)
    all_classes = ('sheila nine stop bed four six down bird marvin cat off right seven eight up three happy go zero '
                   'on wow dog yes five one tree house two left no').split()
    wanted = 'stop down off right up go on yes left no'.split()
    from speech_recognition_amd.model import head32to12          # the product's own map + one kws_head32to12 launch
    p12 = head32to12(p32)
    ref_feat = OF.features(clips, tables, 160, dtype=np.float64).reshape(B, -1)
    ref32 = ora.forward(ref_feat, training=False)
    ref12 = OL.head32to12(ref32, all_classes, wanted)
    assert np.abs(feats.cpu().numpy() - ref_feat).max() < 2e-3
    assert np.abs(p12.cpu().numpy() - ref12).max() < 1e-3          # north_star tolerance on softmax
    assert np.array_equal(p12.cpu().numpy().argmax(1), ref12.argmax(1))
    lib.kws_stft_plan_destroy(plan)


def test_spectrogram_family_on_the_spec_generator(repo_root):
    """conv_1d_spectrogram end to end like train.py would drive it: AudioProcessor(output_representation='spec')
    -> data_gen -> speech_model('conv_1d_spectrogram', fingerprint_size, **model_settings) -> train_on_batch."""
    import sys
    sys.path.insert(0, repo_root)
    import bench
    from speech_recognition_amd.input_data import AudioProcessor, prepare_words_list
    from speech_recognition_amd.model import prepare_model_settings, speech_model
    from speech_recognition_amd.utils import data_gen
    dev = torch.device("cuda", 0)
    settings = prepare_model_settings(label_count=len(prepare_words_list(bench.WANTED)), sample_rate=16000,
                                      clip_duration_ms=1000, window_size_ms=30.0, window_stride_ms=10.0,
                                      dct_coefficient_count=80, num_log_mel_features=60, output_representation='spec')
    assert settings['fingerprint_size'] == 98 * 257
    proc = AudioProcessor(bench.build_synthetic(dev, 8192, seed=59185), 13.0, 60.0, bench.WANTED, 10.0, 0.0, settings,
                          output_representation='spec', device=dev)
    np.random.seed(1234)
    gen = data_gen(proc, None, batch_size=64, mode='training')
    model = speech_model('conv_1d_spectrogram', settings['fingerprint_size'], num_classes=settings['label_count'],
                         **settings)
    assert model.name == 'conv_1d_spectrogram' and abs(float(model.optimizer.lr) - 3e-4) < 1e-9  # f32 variable
    losses = []
    for _ in range(12):
        X, y = next(gen)
        assert np.asarray(X).shape == (64, 98 * 257)
        losses.append(float(model.train_on_batch(X, y)[0]))
    assert np.all(np.isfinite(losses)) and np.mean(losses[-3:]) < np.mean(losses[:3])


@pytest.mark.parametrize("T", [65, 67, 99])
def test_odd_lengths_pool_in_ceil_mode(T):
    """The reference function's own default spectrogram_length = 65 (model.py:1410): 63 frames after the first
    convolution, and Keras' MaxPool1D(2, 2, 'same') / Conv1D(strides=2, 'same') give ceil: 63 -> 32 -> 16 -> 8.  Forward,
    loss and every gradient against the oracle (which is checked against torch's ceil_mode pooling in test_oracle_net)."""
    B, nc, F = 5, 32, 40
    ora, net = _pair(nc, F=F, T=T)
    lens = [T - 2]
    for blk in ora.blocks:
        assert blk['Lin'] == lens[-1] and blk['Lout'] == -(-blk['Lin'] // blk['stride'])
        lens.append(blk['Lout'])
    assert any(blk['Lin'] % 2 for blk in ora.blocks if blk['stride'] == 2)       # a short last window exists
    x, y = _batch(B, nc, T, F, T)
    p_inf = net.predict(torch.from_numpy(x).cuda()).cpu().numpy()
    ref_inf = ora.forward(x.astype(np.float64), training=False)
    assert np.abs(p_inf - ref_inf).max() < 1e-5 and np.array_equal(p_inf.argmax(1), ref_inf.argmax(1))
    probs = net.train_fwd_bwd(torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda(), seed=5, step=1)
    torch.cuda.synchronize()
    masks, args = _decisions(net, ora, B)
    loss, p, grads, cache = ora.loss_and_grads(x.astype(np.float64), y.astype(np.float64), seed=5, step=1,
                                               relu_masks=masks, pool_args=args)
    got = probs.cpu().numpy()
    assert np.abs(got - p).max() < 2e-5 and np.array_equal(got.argmax(1), p.argmax(1))
    assert abs(net.metrics.cpu().numpy()[0] / B - loss) < 5e-5
    g = net.grads_dict()
    for k, ref in grads.items():
        if k in ora.l2_names:
            ref = ref - 2e-5 * ora.params[k].astype(np.float64)
        ref = ref.reshape(g[k].shape)
        assert np.abs(g[k] - ref).max() / max(np.abs(ref).max(), 1e-7) < 1e-4, k


def test_paired_backward_launches_are_bit_identical_for_the_residual_program():
    """Round 4: conv_1d_log_mfcc's 21 input-gradient / weight-gradient pairs go out as single launches (gemm mode 0); mode 1
    makes the two launches of rounds 1 - 3.  Same kernels' code paths: every gradient agrees bit for bit."""
    net = DeviceNet(_lib.KWS_NET_LOG_MFCC, 32, input_size=98 * 40, spectrogram_length=98, num_features=40)
    net.initialize(seed=5)
    w0, s0 = net.params.clone(), net.state.clone()
    g = torch.Generator(device="cuda")
    g.manual_seed(8)
    B = 96
    x = (torch.randn((B, 98 * 40), generator=g, device="cuda") * 3.0).contiguous()
    y = torch.eye(32, device="cuda")[torch.randint(0, 32, (B,), generator=g, device="cuda")].contiguous()
    mode0 = net.gemm_mode
    out = {}
    try:
        for mode in (0, 1):
            net.set_gemm_mode(mode)
            net.params.copy_(w0)
            net.state.copy_(s0)
            p = net.train_fwd_bwd(x, y, seed=3, step=1).clone()
            torch.cuda.synchronize()
            out[mode] = (p, net.grads.clone(), net.metrics.clone(), net.state.clone())
    finally:
        net.set_gemm_mode(mode0)
    for a, b in zip(out[0], out[1]):
        assert torch.equal(a, b)
    assert float(out[0][1].abs().max()) > 0
