"""Oracle: the two Depthwise1D networks on the hot path, forward + backward + step.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

  * TimeSlicedAttentionNet - reference model.py:775-838
    (`conv_1d_time_sliced_with_attention_model`, the model train.py:50-54 builds);
    layer table SURVEY.md Appendix B.1, variable shapes pinned by fixture K1.
  * LogMfccNet - reference model.py:1400-1479 (`conv_1d_log_mfcc_model`), Appendix B.2.

Parameters are kept in an ordered dict under their Keras variable names so the
HIP implementation's flat parameter buffer can be compared tensor by tensor.
"""
from collections import OrderedDict

import numpy as np

from . import layers as L

# (stride, padding, Cout) of the 11 depthwise blocks of model.py:812-817:
# _context_conv(128) then 5 x _reduce_block(n) = [_reduce_conv(n, s2 same), _context_conv(n, s1 valid)]
TS_BLOCKS = [(1, 'valid', 128),
             (2, 'same', 192), (1, 'valid', 192),
             (2, 'same', 256), (1, 'valid', 256),
             (2, 'same', 320), (1, 'valid', 320),
             (2, 'same', 384), (1, 'valid', 384),
             (2, 'same', 512), (1, 'valid', 512)]


def glorot_uniform(rng, shape, fan_in, fan_out):
    """Keras glorot_uniform: U(-l, l), l = sqrt(6/(fan_in+fan_out))."""
    limit = np.sqrt(6.0 / (fan_in + fan_out))
    return rng.uniform(-limit, limit, size=shape).astype(np.float32)


class TimeSlicedAttentionNet(object):
    def __init__(self, num_classes=12, filter_mult=1, input_size=16000, seed=87654321,
                 dtype=np.float64):
        self.dtype = dtype
        self.num_classes = num_classes
        self.input_size = input_size
        rng = np.random.RandomState(seed)
        P = OrderedDict()
        S = OrderedDict()
        c0 = 128 * filter_mult
        P['conv1d_1/kernel'] = glorot_uniform(rng, (3, 40, c0), 3 * 40, 3 * c0)
        self._add_bn(P, S, 1, c0)
        self.blocks = []
        cin = c0
        Lcur = L.valid_len(-(-input_size // 20), 3, 2)
        for i, (stride, padding, cout) in enumerate(TS_BLOCKS):
            cout *= filter_mult
            # DepthwiseConv2D kernel [1,3,C,1]: fan_in = 3*C, fan_out = 3*1
            P['depthwise_conv2d_%d/depthwise_kernel' % (i + 1)] = glorot_uniform(
                rng, (1, 3, cin, 1), 3 * cin, 3)
            P['conv1d_%d/kernel' % (i + 2)] = glorot_uniform(rng, (1, cin, cout), cin, cout)
            self._add_bn(P, S, i + 2, cout)
            if padding == 'same':
                Lout, pl, pr = L.same_pad(Lcur, 3, stride)
            else:
                Lout, pl, pr = L.valid_len(Lcur, 3, stride), 0, 0
            self.blocks.append(dict(stride=stride, pad=(pl, pr), cin=cin, cout=cout, Lin=Lcur, Lout=Lout))
            cin, Lcur = cout, Lout
        self.T, self.C = Lcur, cin
        P['dense_1/kernel'] = glorot_uniform(rng, (self.T * cin, self.T), self.T * cin, self.T)
        P['dense_1/bias'] = np.zeros((self.T,), np.float32)
        P['dense_2/kernel'] = glorot_uniform(rng, (2 * cin, num_classes), 2 * cin, num_classes)
        self.params = P
        self.state = S
        self.l2_names = [k for k in P if k.endswith('kernel')]
        self.drop_keep = 0.6       # Dropout(0.4), model.py:819,828
        self.label_smoothing = 0.1  # model.py:835-836

    @staticmethod
    def _add_bn(P, S, idx, c):
        P['batch_normalization_%d/gamma' % idx] = np.ones((c,), np.float32)
        P['batch_normalization_%d/beta' % idx] = np.zeros((c,), np.float32)
        S['batch_normalization_%d/moving_mean' % idx] = np.zeros((c,), np.float32)
        S['batch_normalization_%d/moving_variance' % idx] = np.ones((c,), np.float32)

    def count_params(self):
        return sum(v.size for v in self.params.values()) + sum(v.size for v in self.state.values())

    # -- helpers ---------------------------------------------------------------
    def _p(self, name):
        return self.params[name].astype(self.dtype)

    def _bn_fwd(self, idx, y, training, cache):
        g = self._p('batch_normalization_%d/gamma' % idx)
        b = self._p('batch_normalization_%d/beta' % idx)
        if training:
            pre, stats = L.bn_train_fwd(y, g, b)
            cache['bn%d' % idx] = (y, g, stats, pre)
            cache.setdefault('batch_stats', OrderedDict())[idx] = (stats[0], stats[1])
        else:
            pre = L.bn_infer_fwd(y, g, b,
                                 self.state['batch_normalization_%d/moving_mean' % idx].astype(self.dtype),
                                 self.state['batch_normalization_%d/moving_variance' % idx].astype(self.dtype))
        return L.relu6(pre)

    def _bn_bwd(self, idx, da, cache, grads):
        y, g, stats, pre = cache['bn%d' % idx]
        mask = L.relu6_mask(pre)
        override = cache.get('relu_masks')
        if override is not None and idx in override:
            # decision-aligned parity: use the mask another implementation actually took (a
            # pre-activation within rounding of a ReLU6 kink may fall on either side in f32 vs f64)
            mask = np.asarray(override[idx], dtype=self.dtype).reshape(pre.shape)
        dpre = da * mask
        dy, dg, db = L.bn_train_bwd(dpre, y, g, stats)
        grads['batch_normalization_%d/gamma' % idx] = dg
        grads['batch_normalization_%d/beta' % idx] = db
        return dy

    # -- forward ---------------------------------------------------------------
    def forward(self, x, training=False, seed=0, step=0, cache=None, drop_offset=0):
        """x [B, input_size] -> softmax probabilities [B, num_classes].
        drop_offset = global index of row 0 (for data-parallel shards)."""
        dt = self.dtype
        cache = {} if cache is None else cache
        x = np.asarray(x, dtype=dt)
        B = x.shape[0]
        frames = L.frame_same(x, 40, 20)                                   # model.py:805
        y, cols = L.conv1d_fwd(frames, self._p('conv1d_1/kernel'), stride=2)   # model.py:807
        cache['conv1_cols'] = cols
        a = self._bn_fwd(1, y, training, cache)
        for i, blk in enumerate(self.blocks):                              # model.py:812-817
            w = self._p('depthwise_conv2d_%d/depthwise_kernel' % (i + 1)).reshape(3, blk['cin'])
            z = L.dwconv_fwd(a, w, blk['stride'], blk['pad'])
            cache['dw%d' % (i + 1)] = (a, w)
            W = self._p('conv1d_%d/kernel' % (i + 2)).reshape(blk['cin'], blk['cout'])
            y = L.pw_fwd(z, W)
            cache['pw%d' % (i + 2)] = (z, W)
            a = self._bn_fwd(i + 2, y, training, cache)
        T, C = self.T, self.C
        flat = a.reshape(B, T * C)                                         # Flatten: index t*C + c
        if training:
            m1 = L.dropout_mask(L.dropout_key(seed, step, 1), B * T * C, self.drop_keep,
                                drop_offset * T * C).reshape(B, T * C)
            fd = flat * m1 / dt(self.drop_keep)
        else:
            m1 = None
            fd = flat
        W1, b1 = self._p('dense_1/kernel'), self._p('dense_1/bias')
        att = L.softmax(fd @ W1 + b1, axis=1)                              # model.py:820-821  [B, T]
        xa = a * att[:, :, None]                                           # model.py:824
        xmax = xa.max(axis=1)                                              # model.py:825
        xavg = a.mean(axis=1)                                              # model.py:826 (unweighted x)
        feat = np.concatenate([xmax, xavg], axis=1)                        # model.py:827
        if training:
            m2 = L.dropout_mask(L.dropout_key(seed, step, 2), B * 2 * C, self.drop_keep,
                                drop_offset * 2 * C).reshape(B, 2 * C)
            featd = feat * m2 / dt(self.drop_keep)
        else:
            m2 = None
            featd = feat
        W2 = self._p('dense_2/kernel')
        p = L.softmax(featd @ W2, axis=1)                                  # model.py:829-830
        cache['tail'] = (a, m1, fd, W1, att, xa, xmax, m2, featd, W2, p)
        return p

    def reg_loss(self):
        return sum(L.L2_COEF * float((self._p(k) ** 2).sum()) for k in self.l2_names)

    # -- backward --------------------------------------------------------------
    def loss_and_grads(self, x, y_onehot, seed=0, step=0, drop_offset=0, loss_scale_B=None,
                       relu_masks=None, pool_ind=None):
        """Returns (data_loss, probs, grads incl. L2 terms, cache).  loss_scale_B: divide the
        data-loss gradient by this batch size instead of the local one (data-parallel mean).
        relu_masks {bn index: 0/1 array} / pool_ind [B,T,C] override the discrete decisions of the
        backward pass (ReLU6 masks, max-pool winners) with those another implementation took."""
        dt = self.dtype
        cache = {'relu_masks': relu_masks}
        p = self.forward(x, training=True, seed=seed, step=step, cache=cache, drop_offset=drop_offset)
        y_onehot = np.asarray(y_onehot, dtype=dt)
        loss, per, dp = L.smooth_cce_fwd_bwd(p, y_onehot, self.label_smoothing)
        B = x.shape[0]
        if loss_scale_B is not None:
            dp = dp * dt(B) / dt(loss_scale_B)
        grads = OrderedDict()
        a, m1, fd, W1, att, xa, xmax, m2, featd, W2, p = cache['tail']
        T, C = self.T, self.C
        dl2 = L.softmax_bwd(dp, p, axis=1)
        grads['dense_2/kernel'] = featd.T @ dl2
        dfeat = (dl2 @ W2.T) * m2 / dt(self.drop_keep)
        dxmax, dxavg = dfeat[:, :C], dfeat[:, C:]
        # reduce_max gradient: split equally among ties (_MinOrMaxGrad)
        ind = (xa == xmax[:, None, :]).astype(dt)
        if pool_ind is not None:
            ind = np.asarray(pool_ind, dtype=dt).reshape(xa.shape)
        ind = ind / ind.sum(axis=1, keepdims=True)
        dxa = ind * dxmax[:, None, :]
        da = dxa * att[:, :, None] + dxavg[:, None, :] / dt(T)
        datt = (dxa * a).sum(axis=2)
        dl1 = L.softmax_bwd(datt, att, axis=1)
        grads['dense_1/kernel'] = fd.T @ dl1
        grads['dense_1/bias'] = dl1.sum(axis=0)
        da = da + ((dl1 @ W1.T) * m1 / dt(self.drop_keep)).reshape(B, T, C)
        for i in reversed(range(len(self.blocks))):
            blk = self.blocks[i]
            dy = self._bn_bwd(i + 2, da, cache, grads)
            z, W = cache['pw%d' % (i + 2)]
            dz, dW = L.pw_bwd(dy, z, W)
            grads['conv1d_%d/kernel' % (i + 2)] = dW.reshape(1, blk['cin'], blk['cout'])
            a_in, w = cache['dw%d' % (i + 1)]
            da, dw = L.dwconv_bwd(dz, a_in, w, blk['stride'], blk['pad'])
            grads['depthwise_conv2d_%d/depthwise_kernel' % (i + 1)] = dw.reshape(1, 3, blk['cin'], 1)
        dy = self._bn_bwd(1, da, cache, grads)
        Wc = self._p('conv1d_1/kernel')
        B2, Lo, Co = dy.shape
        grads['conv1d_1/kernel'] = (cache['conv1_cols'].T @ dy.reshape(B2 * Lo, Co)).reshape(Wc.shape)
        for k in self.l2_names:                                            # kernel_regularizer=l2(1e-5)
            grads[k] = grads[k] + dt(2.0 * L.L2_COEF) * self._p(k)
        ordered = OrderedDict((k, grads[k]) for k in self.params)
        return loss, p, ordered, cache

    # -- training step ---------------------------------------------------------
    def init_optimizer(self, kind='rmsprop'):
        self.opt_kind = kind
        self.slots = OrderedDict((k, np.zeros(v.shape, self.dtype)) for k, v in self.params.items())
        self.master = OrderedDict((k, v.astype(self.dtype)) for k, v in self.params.items())

    def train_step(self, x, y_onehot, lr, seed=0, step=0):
        """One Keras train_on_batch: forward, loss (+L2), backward, optimizer update, BN
        moving-average update.  Returns (total_loss, categorical_accuracy)."""
        loss, p, grads, cache = self.loss_and_grads(x, y_onehot, seed, step)
        total = loss + self.reg_loss()
        for k in self.params:
            if self.opt_kind == 'rmsprop':
                self.master[k], self.slots[k] = L.rmsprop_step(self.master[k], grads[k].reshape(self.master[k].shape),
                                                               self.slots[k], lr)
            else:
                self.master[k], self.slots[k] = L.sgd_momentum_step(self.master[k], grads[k].reshape(self.master[k].shape),
                                                                    self.slots[k], lr)
            self.params[k] = self.master[k].astype(np.float32) if self.dtype == np.float32 else self.master[k]
        for idx, (mean, var) in cache['batch_stats'].items():
            for nm, val in (('moving_mean', mean), ('moving_variance', var)):
                key = 'batch_normalization_%d/%s' % (idx, nm)
                self.state[key] = L.bn_moving_update(self.state[key].astype(self.dtype), val)
        acc = float((p.argmax(axis=1) == np.asarray(y_onehot).argmax(axis=1)).mean())
        return float(total), acc


# ================================================================================================
# a19: conv_1d_log_mfcc_model (reference model.py:1400-1479), SURVEY Appendix B.2
# ================================================================================================
LM_BLOCKS = [(64, 1), (64, 1), (128, 2), (128, 1), (192, 2), (192, 1), (192, 1), (256, 2), (256, 1), (256, 1)]


def maxpool_same_fwd(a, pool):
    """MaxPool1D(pool_size=pool, strides=pool, padding='same') (model.py:1440): ceil(L / pool) windows; TF pads the END
    of a length that is not a multiple of the pool (with -inf for max pooling), so the last window is short."""
    if pool == 1:
        return a, None
    B, L, C = a.shape
    Lo = -(-L // pool)
    if Lo * pool != L:
        a = np.concatenate([a, np.full((B, Lo * pool - L, C), -np.inf, dtype=a.dtype)], axis=1)
    w = a.reshape(B, Lo, pool, C)
    return w.max(axis=2), w.argmax(axis=2)     # first maximum wins (MaxPoolGrad semantics)


def maxpool_same_bwd(do, arg, pool, L):
    if pool == 1:
        return do
    B, Lo, C = do.shape
    d = np.zeros((B, Lo, pool, C), dtype=do.dtype)
    for j in range(pool):
        d[:, :, j, :] = do * (arg == j)
    return d.reshape(B, Lo * pool, C)[:, :L, :]   # the padded positions never win


class LogMfccNet(object):
    """Residual depthwise/pointwise 1-D CNN on [spectrogram_length, num_log_mel_features] features with a
    softmax-over-time attention and global average pooling.  Parameter names follow Keras' per-class
    auto-numbering in layer CREATION order (shortcut Conv1D + BN of a strided block are created before
    the block's depthwise layers, model.py:1429-1441)."""

    def __init__(self, num_classes=32, spectrogram_length=98, num_features=40, seed=87654321, dtype=np.float64):
        self.dtype = dtype
        self.num_classes = num_classes
        self.T0, self.F = spectrogram_length, num_features
        rng = np.random.RandomState(seed)
        P, S = OrderedDict(), OrderedDict()
        self.cnt = dict(conv=0, bn=0, dw=0)
        self.l2_names = []

        def conv(k, cin, cout, l2):
            self.cnt['conv'] += 1
            name = 'conv1d_%d/kernel' % self.cnt['conv']
            P[name] = glorot_uniform(rng, (k, cin, cout), k * cin, k * cout)
            if l2:
                self.l2_names.append(name)
            return name

        def bn(c):
            self.cnt['bn'] += 1
            TimeSlicedAttentionNet._add_bn(P, S, self.cnt['bn'], c)
            return self.cnt['bn']

        def dw(c):
            self.cnt['dw'] += 1
            name = 'depthwise_conv2d_%d/depthwise_kernel' % self.cnt['dw']
            P[name] = glorot_uniform(rng, (1, 3, c, 1), 3 * c, 3)
            self.l2_names.append(name)
            return name

        self.first = (conv(3, num_features, 64, True), bn(64))
        self.blocks = []
        cin, L = 64, spectrogram_length - 2
        if L < 1:
            raise ValueError("LogMfccNet: spectrogram_length %d is too short" % spectrogram_length)
        for nf, stride in LM_BLOCKS:
            # Keras' MaxPool1D(2, 2, 'same') and Conv1D(nf, 1, strides=2, 'same') both give ceil(L / 2): the function's
            # own default spectrogram_length = 65 runs 63 -> 32 -> 16 -> 8 (model.py:1410)
            blk = dict(nf=nf, stride=stride, cin=cin, Lin=L, Lout=-(-L // stride))
            if stride != 1:
                blk['short'] = (conv(1, cin, nf, False), bn(nf))   # no kernel_regularizer (model.py:1431-1432)
            blk['dw1'], blk['pw1'], blk['bn1'] = dw(cin), conv(1, cin, nf, True), bn(nf)
            blk['dw2'], blk['pw2'], blk['bn2'] = dw(nf), conv(1, nf, nf, True), bn(nf)
            self.blocks.append(blk)
            cin, L = nf, -(-L // stride)
        self.T, self.C = L, cin
        self.att = (dw(cin), conv(1, cin, 1, True), bn(1))         # _context_conv(x, 1, 3, 'same'), model.py:1464
        P['dense_1/kernel'] = glorot_uniform(rng, (cin, num_classes), cin, num_classes)
        P['dense_1/bias'] = np.zeros((num_classes,), np.float32)
        self.l2_names.append('dense_1/kernel')
        self.params, self.state = P, S
        self.drop_keep = 0.8                                       # Dropout(0.2), model.py:1471

    def count_params(self):
        return sum(v.size for v in self.params.values()) + sum(v.size for v in self.state.values())

    def _p(self, name):
        return self.params[name].astype(self.dtype)

    def _bn(self, idx, y, training, cache, relu=True):
        g = self._p('batch_normalization_%d/gamma' % idx)
        b = self._p('batch_normalization_%d/beta' % idx)
        if training:
            pre, stats = L.bn_train_fwd(y, g, b)
            cache['bn%d' % idx] = (y, g, stats, pre)
            cache.setdefault('batch_stats', OrderedDict())[idx] = (stats[0], stats[1])
        else:
            pre = L.bn_infer_fwd(y, g, b,
                                 self.state['batch_normalization_%d/moving_mean' % idx].astype(self.dtype),
                                 self.state['batch_normalization_%d/moving_variance' % idx].astype(self.dtype))
        return L.relu6(pre) if relu else pre

    def _bn_bwd(self, idx, dout, cache, grads, relu=True):
        y, g, stats, pre = cache['bn%d' % idx]
        if relu:
            mask = L.relu6_mask(pre)
            ov = cache.get('relu_masks')
            if ov is not None and idx in ov:
                mask = np.asarray(ov[idx], dtype=self.dtype).reshape(pre.shape)
            dout = dout * mask
        dy, dg, db = L.bn_train_bwd(dout, y, g, stats)
        grads['batch_normalization_%d/gamma' % idx] = dg
        grads['batch_normalization_%d/beta' % idx] = db
        return dy

    def forward(self, x, training=False, seed=0, step=0, cache=None, drop_offset=0):
        dt = self.dtype
        cache = {} if cache is None else cache
        B = x.shape[0]
        h = np.asarray(x, dtype=dt).reshape(B, self.T0, self.F)                       # Reshape, model.py:1446
        y, cols = L.conv1d_fwd(h, self._p(self.first[0]), stride=1)                   # Conv1D(64,3) valid
        cache['conv1_cols'] = cols
        h = self._bn(self.first[1], y, training, cache)
        for i, blk in enumerate(self.blocks):                                         # model.py:1453-1462
            c = {}
            c['x'] = h
            if 'short' in blk:
                xs = h[:, ::blk['stride'], :]                                         # Conv1D(nf,1,strides,same)
                Ws = self._p(blk['short'][0]).reshape(blk['cin'], blk['nf'])
                c['xs'], c['Ws'] = xs, Ws
                res = self._bn(blk['short'][1], L.pw_fwd(xs, Ws), training, cache, relu=False)
            else:
                res = h
            w1 = self._p(blk['dw1']).reshape(3, blk['cin'])
            z1 = L.dwconv_fwd(h, w1, 1, (1, 1))
            W1 = self._p(blk['pw1']).reshape(blk['cin'], blk['nf'])
            a1 = self._bn(blk['bn1'], L.pw_fwd(z1, W1), training, cache)
            w2 = self._p(blk['dw2']).reshape(3, blk['nf'])
            z2 = L.dwconv_fwd(a1, w2, 1, (1, 1))
            W2 = self._p(blk['pw2']).reshape(blk['nf'], blk['nf'])
            a2 = self._bn(blk['bn2'], L.pw_fwd(z2, W2), training, cache)
            pooled, arg = maxpool_same_fwd(a2, blk['stride'])
            c.update(w1=w1, z1=z1, W1=W1, a1=a1, w2=w2, z2=z2, W2=W2, arg=arg)
            cache['blk%d' % i] = c
            h = pooled + res                                                          # Add
        wa = self._p(self.att[0]).reshape(3, self.C)
        za = L.dwconv_fwd(h, wa, 1, (1, 1))
        Wa = self._p(self.att[1]).reshape(self.C, 1)
        u = self._bn(self.att[2], L.pw_fwd(za, Wa), training, cache)                  # [B, T, 1]
        att = L.softmax(u, axis=1)                                                    # softmax over time
        feat = (h * att).mean(axis=1)                                                 # Multiply + GAP
        if training:
            m = L.dropout_mask(L.dropout_key(seed, step, 1), B * self.C, self.drop_keep,
                               drop_offset * self.C).reshape(B, self.C)
            fd = feat * m / dt(self.drop_keep)
        else:
            m, fd = None, feat
        Wd, bd = self._p('dense_1/kernel'), self._p('dense_1/bias')
        p = L.softmax(fd @ Wd + bd, axis=1)
        cache['tail'] = (h, wa, za, Wa, u, att, m, fd, Wd, p)
        return p

    def reg_loss(self):
        return sum(L.L2_COEF * float((self._p(k) ** 2).sum()) for k in self.l2_names)

    def loss_and_grads(self, x, y_onehot, seed=0, step=0, drop_offset=0, loss_scale_B=None, relu_masks=None,
                       pool_args=None):
        dt = self.dtype
        cache = {'relu_masks': relu_masks}
        p = self.forward(x, training=True, seed=seed, step=step, cache=cache, drop_offset=drop_offset)
        y_onehot = np.asarray(y_onehot, dtype=dt)
        loss, per, dp = L.cce_fwd_bwd(p, y_onehot)                                    # model.py:1477
        B = x.shape[0]
        if loss_scale_B is not None:
            dp = dp * dt(B) / dt(loss_scale_B)
        grads = OrderedDict()
        h, wa, za, Wa, u, att, m, fd, Wd, p = cache['tail']
        dl = L.softmax_bwd(dp, p, axis=1)
        grads['dense_1/kernel'] = fd.T @ dl
        grads['dense_1/bias'] = dl.sum(axis=0)
        dfeat = (dl @ Wd.T) * m / dt(self.drop_keep)
        dprod = np.repeat(dfeat[:, None, :], self.T, axis=1) / dt(self.T)             # GAP backward
        dh = dprod * att
        datt = (dprod * h).sum(axis=2, keepdims=True)
        du = L.softmax_bwd(datt, att, axis=1)
        dyu = self._bn_bwd(self.att[2], du, cache, grads)
        dza, dWa = L.pw_bwd(dyu, za, Wa)
        grads[self.att[1]] = dWa.reshape(1, self.C, 1)
        dh2, dwa = L.dwconv_bwd(dza, h, wa, 1, (1, 1))
        grads[self.att[0]] = dwa.reshape(1, 3, self.C, 1)
        dh = dh + dh2
        for i in reversed(range(len(self.blocks))):
            blk, c = self.blocks[i], cache['blk%d' % i]
            arg = c['arg'] if pool_args is None or i not in pool_args else pool_args[i]
            da2 = maxpool_same_bwd(dh, arg, blk['stride'], blk['Lin'])
            dy2 = self._bn_bwd(blk['bn2'], da2, cache, grads)
            dz2, dW2 = L.pw_bwd(dy2, c['z2'], c['W2'])
            grads[blk['pw2']] = dW2.reshape(1, blk['nf'], blk['nf'])
            da1, dw2 = L.dwconv_bwd(dz2, c['a1'], c['w2'], 1, (1, 1))
            grads[blk['dw2']] = dw2.reshape(1, 3, blk['nf'], 1)
            dy1 = self._bn_bwd(blk['bn1'], da1, cache, grads)
            dz1, dW1 = L.pw_bwd(dy1, c['z1'], c['W1'])
            grads[blk['pw1']] = dW1.reshape(1, blk['cin'], blk['nf'])
            dx, dw1 = L.dwconv_bwd(dz1, c['x'], c['w1'], 1, (1, 1))
            grads[blk['dw1']] = dw1.reshape(1, 3, blk['cin'], 1)
            if 'short' in blk:
                dys = self._bn_bwd(blk['short'][1], dh, cache, grads, relu=False)
                dxs, dWs = L.pw_bwd(dys, c['xs'], c['Ws'])
                grads[blk['short'][0]] = dWs.reshape(1, blk['cin'], blk['nf'])
                dx = dx.copy()
                dx[:, ::blk['stride'], :] += dxs
            else:
                dx = dx + dh
            dh = dx
        dy = self._bn_bwd(self.first[1], dh, cache, grads)
        Wc = self._p(self.first[0])
        B2, Lo, Co = dy.shape
        grads[self.first[0]] = (cache['conv1_cols'].T @ dy.reshape(B2 * Lo, Co)).reshape(Wc.shape)
        for k in self.l2_names:
            grads[k] = grads[k] + dt(2.0 * L.L2_COEF) * self._p(k)
        return loss, p, OrderedDict((k, grads[k]) for k in self.params), cache


# ----------------------------------------------------------------------------------------------------------
# steffeNet (reference model.py:1663-1726)
# ----------------------------------------------------------------------------------------------------------
STEFFE_WIDTHS = [320, 384, 512, 768, 1024, 1536]                   # model.py:1709


class SteffeNet(LogMfccNet):
    """Raw waveform -> Conv1D(256, 75, strides=50, same, no bias, no regulariser) + BN + ReLU6 ->
    _context_conv(256, 3, same) -> 6 x [residual block stride 2, residual block stride 1] ->
    GlobalMaxPooling1D ++ GlobalAveragePooling1D -> Dropout(.5) -> Dense(num_classes, no bias) + softmax,
    label-smoothed CE (0.1), RMSprop(1e-3).

    A residual block (model.py:1690-1701) differs from conv_1d_log_mfcc's in where the stride sits: the FIRST
    depthwise convolution is strided (SAME), there is no max-pool, and the sum is not activated.  Keras names in
    layer creation order, as in LogMfccNet (shortcut Conv1D + BN first)."""

    def __init__(self, num_classes=12, input_size=16000, filter_widths=STEFFE_WIDTHS, c0=256, seed=87654321,
                 dtype=np.float64):
        self.dtype = dtype
        self.num_classes = num_classes
        self.L_in = input_size
        rng = np.random.RandomState(seed)
        P, S = OrderedDict(), OrderedDict()
        self.cnt = dict(conv=0, bn=0, dw=0)
        self.l2_names = []

        def conv(k, cin, cout, l2):
            self.cnt['conv'] += 1
            name = 'conv1d_%d/kernel' % self.cnt['conv']
            P[name] = glorot_uniform(rng, (k, cin, cout), k * cin, k * cout)
            if l2:
                self.l2_names.append(name)
            return name

        def bn(c):
            self.cnt['bn'] += 1
            TimeSlicedAttentionNet._add_bn(P, S, self.cnt['bn'], c)
            return self.cnt['bn']

        def dw(c):
            self.cnt['dw'] += 1
            name = 'depthwise_conv2d_%d/depthwise_kernel' % self.cnt['dw']
            P[name] = glorot_uniform(rng, (1, 3, c, 1), 3 * c, 3)
            self.l2_names.append(name)
            return name

        self.K0, self.S0, self.C0 = 75, 50, c0
        self.L0, self.pl0, self.pr0 = L.same_pad(input_size, self.K0, self.S0)
        self.first = (conv(self.K0, 1, c0, False), bn(c0))                     # model.py:1705-1707
        self.ctx = (dw(c0), conv(1, c0, c0, True), bn(c0))                     # _context_conv(x, 256, 3, 'same')
        self.blocks = []
        cin, Lc = c0, self.L0
        for nh in filter_widths:
            for stride in (2, 1):
                Lout, pl, pr = L.same_pad(Lc, 3, stride)
                blk = dict(nf=nh, stride=stride, cin=cin, Lin=Lc, Lout=Lout, pad1=(pl, pr))
                if stride != 1:
                    blk['short'] = (conv(1, cin, nh, False), bn(nh))
                blk['dw1'], blk['pw1'], blk['bn1'] = dw(cin), conv(1, cin, nh, True), bn(nh)
                blk['dw2'], blk['pw2'], blk['bn2'] = dw(nh), conv(1, nh, nh, True), bn(nh)
                self.blocks.append(blk)
                cin, Lc = nh, Lout
        self.T, self.C = Lc, cin
        P['dense_1/kernel'] = glorot_uniform(rng, (2 * cin, num_classes), 2 * cin, num_classes)
        self.l2_names.append('dense_1/kernel')
        self.params, self.state = P, S
        self.drop_keep = 0.5                                                   # Dropout(0.5), model.py:1716
        self.label_smoothing = 0.1                                             # model.py:1722-1724

    def forward(self, x, training=False, seed=0, step=0, cache=None, drop_offset=0):
        dt = self.dtype
        cache = {} if cache is None else cache
        B = x.shape[0]
        h = np.asarray(x, dtype=dt).reshape(B, self.L_in, 1)                   # Reshape([-1, 1])
        y, cols = L.conv1d_fwd(h, self._p(self.first[0]), stride=self.S0, pad=(self.pl0, self.pr0))
        cache['conv1_cols'] = cols
        h = self._bn(self.first[1], y, training, cache)
        wc = self._p(self.ctx[0]).reshape(3, self.C0)
        zc = L.dwconv_fwd(h, wc, 1, (1, 1))
        Wc = self._p(self.ctx[1]).reshape(self.C0, self.C0)
        cache['ctx'] = (h, wc, zc, Wc)
        h = self._bn(self.ctx[2], L.pw_fwd(zc, Wc), training, cache)
        for i, blk in enumerate(self.blocks):
            c = {'x': h}
            if 'short' in blk:
                xs = h[:, ::blk['stride'], :]                                  # Conv1D(nh, 1, strides, same)
                Ws = self._p(blk['short'][0]).reshape(blk['cin'], blk['nf'])
                c['xs'], c['Ws'] = xs, Ws
                res = self._bn(blk['short'][1], L.pw_fwd(xs, Ws), training, cache, relu=False)
            else:
                res = h
            w1 = self._p(blk['dw1']).reshape(3, blk['cin'])
            z1 = L.dwconv_fwd(h, w1, blk['stride'], blk['pad1'])
            W1 = self._p(blk['pw1']).reshape(blk['cin'], blk['nf'])
            a1 = self._bn(blk['bn1'], L.pw_fwd(z1, W1), training, cache)
            w2 = self._p(blk['dw2']).reshape(3, blk['nf'])
            z2 = L.dwconv_fwd(a1, w2, 1, (1, 1))
            W2 = self._p(blk['pw2']).reshape(blk['nf'], blk['nf'])
            a2 = self._bn(blk['bn2'], L.pw_fwd(z2, W2), training, cache)
            c.update(w1=w1, z1=z1, W1=W1, a1=a1, w2=w2, z2=z2, W2=W2)
            cache['blk%d' % i] = c
            h = a2 + res                                                       # Add, no activation
        xmax, xavg = h.max(axis=1), h.mean(axis=1)
        feat = np.concatenate([xmax, xavg], axis=1)                            # Concatenate()([x_max, x_avg])
        if training:
            m = L.dropout_mask(L.dropout_key(seed, step, 1), B * 2 * self.C, self.drop_keep,
                               drop_offset * 2 * self.C).reshape(B, 2 * self.C)
            fd = feat * m / dt(self.drop_keep)
        else:
            m, fd = None, feat
        Wd = self._p('dense_1/kernel')
        p = L.softmax(fd @ Wd, axis=1)
        cache['tail'] = (h, xmax, m, fd, Wd, p)
        return p

    def loss_and_grads(self, x, y_onehot, seed=0, step=0, drop_offset=0, loss_scale_B=None, relu_masks=None,
                       pool_ind=None):
        dt = self.dtype
        cache = {'relu_masks': relu_masks}
        p = self.forward(x, training=True, seed=seed, step=step, cache=cache, drop_offset=drop_offset)
        y_onehot = np.asarray(y_onehot, dtype=dt)
        loss, per, dp = L.smooth_cce_fwd_bwd(p, y_onehot, self.label_smoothing)
        B = x.shape[0]
        if loss_scale_B is not None:
            dp = dp * dt(B) / dt(loss_scale_B)
        grads = OrderedDict()
        h, xmax, m, fd, Wd, p = cache['tail']
        dl = L.softmax_bwd(dp, p, axis=1)
        grads['dense_1/kernel'] = fd.T @ dl
        dfeat = (dl @ Wd.T) * m / dt(self.drop_keep)
        dxmax, dxavg = dfeat[:, :self.C], dfeat[:, self.C:]
        ind = (h == xmax[:, None, :]).astype(dt)                               # reduce_max: ties share the gradient
        if pool_ind is not None:
            ind = np.asarray(pool_ind, dtype=dt).reshape(h.shape)
        ind = ind / ind.sum(axis=1, keepdims=True)
        dh = ind * dxmax[:, None, :] + dxavg[:, None, :] / dt(self.T)
        for i in reversed(range(len(self.blocks))):
            blk, c = self.blocks[i], cache['blk%d' % i]
            dy2 = self._bn_bwd(blk['bn2'], dh, cache, grads)
            dz2, dW2 = L.pw_bwd(dy2, c['z2'], c['W2'])
            grads[blk['pw2']] = dW2.reshape(1, blk['nf'], blk['nf'])
            da1, dw2 = L.dwconv_bwd(dz2, c['a1'], c['w2'], 1, (1, 1))
            grads[blk['dw2']] = dw2.reshape(1, 3, blk['nf'], 1)
            dy1 = self._bn_bwd(blk['bn1'], da1, cache, grads)
            dz1, dW1 = L.pw_bwd(dy1, c['z1'], c['W1'])
            grads[blk['pw1']] = dW1.reshape(1, blk['cin'], blk['nf'])
            dx, dw1 = L.dwconv_bwd(dz1, c['x'], c['w1'], blk['stride'], blk['pad1'])
            grads[blk['dw1']] = dw1.reshape(1, 3, blk['cin'], 1)
            if 'short' in blk:
                dys = self._bn_bwd(blk['short'][1], dh, cache, grads, relu=False)
                dxs, dWs = L.pw_bwd(dys, c['xs'], c['Ws'])
                grads[blk['short'][0]] = dWs.reshape(1, blk['cin'], blk['nf'])
                dx = dx.copy()
                dx[:, ::blk['stride'], :] += dxs
            else:
                dx = dx + dh
            dh = dx
        hc, wc, zc, Wc = cache['ctx']
        dyc = self._bn_bwd(self.ctx[2], dh, cache, grads)
        dzc, dWc = L.pw_bwd(dyc, zc, Wc)
        grads[self.ctx[1]] = dWc.reshape(1, self.C0, self.C0)
        dh0, dwc = L.dwconv_bwd(dzc, hc, wc, 1, (1, 1))
        grads[self.ctx[0]] = dwc.reshape(1, 3, self.C0, 1)
        dy = self._bn_bwd(self.first[1], dh0, cache, grads)
        W0 = self._p(self.first[0])
        B2, Lo, Co = dy.shape
        grads[self.first[0]] = (cache['conv1_cols'].T @ dy.reshape(B2 * Lo, Co)).reshape(W0.shape)
        for k in self.l2_names:
            grads[k] = grads[k] + dt(2.0 * L.L2_COEF) * self._p(k)
        return loss, p, OrderedDict((k, grads[k]) for k in self.params), cache


# ----------------------------------------------------------------------------------------------------------
# conv_1d_residual_model (reference model.py:841-908)
# ----------------------------------------------------------------------------------------------------------
RES_BLOCKS = [(128, 2), (256, 2)] + [(256, 1)] * 8 + [(512, 2), (728, 2), (728, 2)]    # model.py:888-894


def maxpool3_same_fwd(a, stride):
    """MaxPool1D(pool_size=3, strides=stride, padding='same') on [B, L, C]: -inf padding (TF pads max-pool windows
    with the lowest value), the FIRST maximum of a window wins (MaxPoolGrad's strict '>').  Returns (out, arg)
    with arg in {0, 1, 2} = winner's offset inside its window."""
    B, Lin, C = a.shape
    Lout, pl, pr = L.same_pad(Lin, 3, stride)
    ap = np.pad(a, [[0, 0], [pl, pr], [0, 0]], constant_values=-np.inf)
    win = np.stack([ap[:, j:j + stride * Lout:stride, :] for j in range(3)], axis=2)     # [B, Lout, 3, C]
    return win.max(axis=2), win.argmax(axis=2)


def maxpool3_same_bwd(do, arg, stride, Lin):
    B, Lout, C = do.shape
    _, pl, pr = L.same_pad(Lin, 3, stride)
    dp = np.zeros((B, Lin + pl + pr, C), dtype=do.dtype)
    for j in range(3):
        dp[:, j:j + stride * Lout:stride, :] += do * (arg == j)
    return dp[:, pl:pl + Lin, :]


class Conv1dResidualNet(LogMfccNet):
    """Raw waveform -> overlapping_time_slice_stack(40, 20) -> Conv1D(64, 3, strides=2) + BN + ReLU6 -> 13 residual
    blocks of 2 x [depthwise k3 SAME -> pointwise -> BN -> ReLU6] + MaxPool1D(3, strides, 'same') + Add (1x1 strided
    Conv1D + BN shortcut on the strided blocks) -> _reduce_block(1024) = strided SAME block + VALID block ->
    GlobalAveragePooling1D -> Dropout(.5) -> Dense(num_classes) + softmax, categorical CE, RMSprop(1e-4)."""

    def __init__(self, num_classes=12, input_size=16000, blocks=RES_BLOCKS, c0=64, c_reduce=1024, seed=87654321,
                 dtype=np.float64):
        self.dtype = dtype
        self.num_classes = num_classes
        self.L_in = input_size
        rng = np.random.RandomState(seed)
        P, S = OrderedDict(), OrderedDict()
        self.cnt = dict(conv=0, bn=0, dw=0)
        self.l2_names = []

        def conv(k, cin, cout, l2):
            self.cnt['conv'] += 1
            name = 'conv1d_%d/kernel' % self.cnt['conv']
            P[name] = glorot_uniform(rng, (k, cin, cout), k * cin, k * cout)
            if l2:
                self.l2_names.append(name)
            return name

        def bn(c):
            self.cnt['bn'] += 1
            TimeSlicedAttentionNet._add_bn(P, S, self.cnt['bn'], c)
            return self.cnt['bn']

        def dw(c):
            self.cnt['dw'] += 1
            name = 'depthwise_conv2d_%d/depthwise_kernel' % self.cnt['dw']
            P[name] = glorot_uniform(rng, (1, 3, c, 1), 3 * c, 3)
            self.l2_names.append(name)
            return name

        self.C0 = c0
        Lf = L.same_pad(input_size, 40, 20)[0]
        self.L0 = L.valid_len(Lf, 3, 2)
        self.first = (conv(3, 40, c0, True), bn(c0))                           # model.py:883-886
        self.blocks = []
        cin, Lc = c0, self.L0
        for nf, stride in blocks:
            Lout = L.same_pad(Lc, 3, stride)[0]
            blk = dict(nf=nf, stride=stride, cin=cin, Lin=Lc, Lout=Lout)
            if stride != 1:
                blk['short'] = (conv(1, cin, nf, False), bn(nf))
            blk['dw1'], blk['pw1'], blk['bn1'] = dw(cin), conv(1, cin, nf, True), bn(nf)
            blk['dw2'], blk['pw2'], blk['bn2'] = dw(nf), conv(1, nf, nf, True), bn(nf)
            self.blocks.append(blk)
            cin, Lc = nf, Lout
        # _reduce_block(x, 1024, 3): _reduce_conv (strides 2, 'same') then _context_conv ('valid')
        Lr, plr, prr = L.same_pad(Lc, 3, 2)
        self.red = [dict(dw=dw(cin), pw=conv(1, cin, c_reduce, True), bn=bn(c_reduce), cin=cin, cout=c_reduce, stride=2,
                         pad=(plr, prr), Lin=Lc, Lout=Lr),
                    dict(dw=dw(c_reduce), pw=conv(1, c_reduce, c_reduce, True), bn=bn(c_reduce), cin=c_reduce,
                         cout=c_reduce, stride=1, pad=(0, 0), Lin=Lr, Lout=Lr - 2)]
        self.T, self.C = Lr - 2, c_reduce
        P['dense_1/kernel'] = glorot_uniform(rng, (c_reduce, num_classes), c_reduce, num_classes)
        P['dense_1/bias'] = np.zeros((num_classes,), np.float32)
        self.l2_names.append('dense_1/kernel')
        self.params, self.state = P, S
        self.drop_keep = 0.5                                                   # Dropout(0.5), model.py:899

    def forward(self, x, training=False, seed=0, step=0, cache=None, drop_offset=0):
        dt = self.dtype
        cache = {} if cache is None else cache
        x = np.asarray(x, dtype=dt)
        B = x.shape[0]
        frames = L.frame_same(x, 40, 20)                                       # model.py:881
        y, cols = L.conv1d_fwd(frames, self._p(self.first[0]), stride=2)
        cache['conv1_cols'] = cols
        h = self._bn(self.first[1], y, training, cache)
        for i, blk in enumerate(self.blocks):
            c = {'x': h}
            if 'short' in blk:
                xs = h[:, ::blk['stride'], :]
                Ws = self._p(blk['short'][0]).reshape(blk['cin'], blk['nf'])
                c['xs'], c['Ws'] = xs, Ws
                res = self._bn(blk['short'][1], L.pw_fwd(xs, Ws), training, cache, relu=False)
            else:
                res = h
            w1 = self._p(blk['dw1']).reshape(3, blk['cin'])
            z1 = L.dwconv_fwd(h, w1, 1, (1, 1))
            W1 = self._p(blk['pw1']).reshape(blk['cin'], blk['nf'])
            a1 = self._bn(blk['bn1'], L.pw_fwd(z1, W1), training, cache)
            w2 = self._p(blk['dw2']).reshape(3, blk['nf'])
            z2 = L.dwconv_fwd(a1, w2, 1, (1, 1))
            W2 = self._p(blk['pw2']).reshape(blk['nf'], blk['nf'])
            a2 = self._bn(blk['bn2'], L.pw_fwd(z2, W2), training, cache)
            pooled, arg = maxpool3_same_fwd(a2, blk['stride'])
            c.update(w1=w1, z1=z1, W1=W1, a1=a1, w2=w2, z2=z2, W2=W2, arg=arg)
            cache['blk%d' % i] = c
            h = pooled + res
        for j, r in enumerate(self.red):
            w = self._p(r['dw']).reshape(3, r['cin'])
            z = L.dwconv_fwd(h, w, r['stride'], r['pad'])
            W = self._p(r['pw']).reshape(r['cin'], r['cout'])
            cache['red%d' % j] = (h, w, z, W)
            h = self._bn(r['bn'], L.pw_fwd(z, W), training, cache)
        feat = h.mean(axis=1)
        if training:
            m = L.dropout_mask(L.dropout_key(seed, step, 1), B * self.C, self.drop_keep,
                               drop_offset * self.C).reshape(B, self.C)
            fd = feat * m / dt(self.drop_keep)
        else:
            m, fd = None, feat
        Wd, bd = self._p('dense_1/kernel'), self._p('dense_1/bias')
        p = L.softmax(fd @ Wd + bd, axis=1)
        cache['tail'] = (m, fd, Wd, p)
        return p

    def loss_and_grads(self, x, y_onehot, seed=0, step=0, drop_offset=0, loss_scale_B=None, relu_masks=None,
                       pool_args=None):
        dt = self.dtype
        cache = {'relu_masks': relu_masks}
        p = self.forward(x, training=True, seed=seed, step=step, cache=cache, drop_offset=drop_offset)
        y_onehot = np.asarray(y_onehot, dtype=dt)
        loss, per, dp = L.cce_fwd_bwd(p, y_onehot)                             # model.py:905
        B = x.shape[0]
        if loss_scale_B is not None:
            dp = dp * dt(B) / dt(loss_scale_B)
        grads = OrderedDict()
        m, fd, Wd, p = cache['tail']
        dl = L.softmax_bwd(dp, p, axis=1)
        grads['dense_1/kernel'] = fd.T @ dl
        grads['dense_1/bias'] = dl.sum(axis=0)
        dfeat = (dl @ Wd.T) * m / dt(self.drop_keep)
        dh = np.repeat(dfeat[:, None, :], self.T, axis=1) / dt(self.T)
        for j in reversed(range(len(self.red))):
            r = self.red[j]
            hin, w, z, W = cache['red%d' % j]
            dy = self._bn_bwd(r['bn'], dh, cache, grads)
            dz, dW = L.pw_bwd(dy, z, W)
            grads[r['pw']] = dW.reshape(1, r['cin'], r['cout'])
            dh, dwk = L.dwconv_bwd(dz, hin, w, r['stride'], r['pad'])
            grads[r['dw']] = dwk.reshape(1, 3, r['cin'], 1)
        for i in reversed(range(len(self.blocks))):
            blk, c = self.blocks[i], cache['blk%d' % i]
            arg = c['arg'] if pool_args is None or i not in pool_args else pool_args[i]
            da2 = maxpool3_same_bwd(dh, arg, blk['stride'], blk['Lin'])
            dy2 = self._bn_bwd(blk['bn2'], da2, cache, grads)
            dz2, dW2 = L.pw_bwd(dy2, c['z2'], c['W2'])
            grads[blk['pw2']] = dW2.reshape(1, blk['nf'], blk['nf'])
            da1, dw2 = L.dwconv_bwd(dz2, c['a1'], c['w2'], 1, (1, 1))
            grads[blk['dw2']] = dw2.reshape(1, 3, blk['nf'], 1)
            dy1 = self._bn_bwd(blk['bn1'], da1, cache, grads)
            dz1, dW1 = L.pw_bwd(dy1, c['z1'], c['W1'])
            grads[blk['pw1']] = dW1.reshape(1, blk['cin'], blk['nf'])
            dx, dw1 = L.dwconv_bwd(dz1, c['x'], c['w1'], 1, (1, 1))
            grads[blk['dw1']] = dw1.reshape(1, 3, blk['cin'], 1)
            if 'short' in blk:
                dys = self._bn_bwd(blk['short'][1], dh, cache, grads, relu=False)
                dxs, dWs = L.pw_bwd(dys, c['xs'], c['Ws'])
                grads[blk['short'][0]] = dWs.reshape(1, blk['cin'], blk['nf'])
                dx = dx.copy()
                dx[:, ::blk['stride'], :] += dxs
            else:
                dx = dx + dh
            dh = dx
        dy = self._bn_bwd(self.first[1], dh, cache, grads)
        Wc = self._p(self.first[0])
        B2, Lo, Co = dy.shape
        grads[self.first[0]] = (cache['conv1_cols'].T @ dy.reshape(B2 * Lo, Co)).reshape(Wc.shape)
        for k in self.l2_names:
            grads[k] = grads[k] + dt(2.0 * L.L2_COEF) * self._p(k)
        return loss, p, OrderedDict((k, grads[k]) for k in self.params), cache


# ----------------------------------------------------------------------------------------------------------
# conv_1d_mfcc_and_raw_model (reference model.py:1563-1660)
# ----------------------------------------------------------------------------------------------------------
MR_BLOCKS = [(160, 1), (160, 1), (192, 2), (192, 1), (256, 2), (256, 1), (320, 2), (320, 1), (384, 2), (384, 1)]


class MfccAndRawNet(Conv1dResidualNet):
    """Two inputs (the generator's 'mfcc_and_raw' output): log-mel features [T, F] -> Conv1D(64, 3) + BN + ReLU6 and raw
    waveform -> overlapping_time_slice_stack(frame_length, frame_step, 'VALID') -> Conv1D(96, 3) + BN + ReLU6,
    concatenated to 160 channels -> 10 residual blocks with MaxPool1D(3, strides, 'same') joins ->
    GlobalAveragePooling1D -> Dropout(.3) -> Dense + softmax, categorical CE, RMSprop(5e-4).
    Keras creation order: mfcc Conv1D + BN, raw Conv1D + BN, then the blocks (model.py:1611-1639)."""

    def __init__(self, num_classes=12, spectrogram_length=98, num_features=60, raw_size=16000, frame_length=480,
                 frame_step=160, blocks=MR_BLOCKS, c_mfcc=64, c_raw=96, seed=87654321, dtype=np.float64):
        self.dtype = dtype
        self.num_classes = num_classes
        self.T0, self.F, self.L_in = spectrogram_length, num_features, raw_size
        self.frame_length, self.frame_step = frame_length, frame_step
        rng = np.random.RandomState(seed)
        P, S = OrderedDict(), OrderedDict()
        self.cnt = dict(conv=0, bn=0, dw=0)
        self.l2_names = []

        def conv(k, cin, cout, l2):
            self.cnt['conv'] += 1
            name = 'conv1d_%d/kernel' % self.cnt['conv']
            P[name] = glorot_uniform(rng, (k, cin, cout), k * cin, k * cout)
            if l2:
                self.l2_names.append(name)
            return name

        def bn(c):
            self.cnt['bn'] += 1
            TimeSlicedAttentionNet._add_bn(P, S, self.cnt['bn'], c)
            return self.cnt['bn']

        def dw(c):
            self.cnt['dw'] += 1
            name = 'depthwise_conv2d_%d/depthwise_kernel' % self.cnt['dw']
            P[name] = glorot_uniform(rng, (1, 3, c, 1), 3 * c, 3)
            self.l2_names.append(name)
            return name

        n_frames = 1 + (raw_size - frame_length) // frame_step            # extract_image_patches VALID
        assert n_frames == spectrogram_length, "both branches must have the same number of time steps"
        self.L0 = spectrogram_length - 2
        self.Cm, self.Cr = c_mfcc, c_raw
        self.first_m = (conv(3, num_features, c_mfcc, True), bn(c_mfcc))  # model.py:1615-1618
        self.first_r = (conv(3, frame_length, c_raw, True), bn(c_raw))    # model.py:1625-1628
        self.blocks = []
        cin, Lc = c_mfcc + c_raw, self.L0
        for nf, stride in blocks:
            Lout = L.same_pad(Lc, 3, stride)[0]
            blk = dict(nf=nf, stride=stride, cin=cin, Lin=Lc, Lout=Lout)
            if stride != 1:
                blk['short'] = (conv(1, cin, nf, False), bn(nf))
            blk['dw1'], blk['pw1'], blk['bn1'] = dw(cin), conv(1, cin, nf, True), bn(nf)
            blk['dw2'], blk['pw2'], blk['bn2'] = dw(nf), conv(1, nf, nf, True), bn(nf)
            self.blocks.append(blk)
            cin, Lc = nf, Lout
        self.red = []
        self.T, self.C = Lc, cin
        P['dense_1/kernel'] = glorot_uniform(rng, (cin, num_classes), cin, num_classes)
        P['dense_1/bias'] = np.zeros((num_classes,), np.float32)
        self.l2_names.append('dense_1/kernel')
        self.params, self.state = P, S
        self.drop_keep = 0.7                                              # Dropout(0.3), model.py:1648

    def _stem(self, x, training, cache):
        """x = [mfcc [B, T*F], raw [B, L]] -> concatenated, activated [B, T-2, 160]."""
        dt = self.dtype
        xm, xr = x
        B = np.asarray(xm).shape[0]
        hm = np.asarray(xm, dtype=dt).reshape(B, self.T0, self.F)
        ym, cols_m = L.conv1d_fwd(hm, self._p(self.first_m[0]), stride=1)
        xr = np.asarray(xr, dtype=dt)
        idx = self.frame_step * np.arange(self.T0)[:, None] + np.arange(self.frame_length)[None, :]
        frames = xr[:, idx]                                               # [B, T, frame_length], VALID
        yr, cols_r = L.conv1d_fwd(frames, self._p(self.first_r[0]), stride=1)
        cache['conv1_cols'] = (cols_m, cols_r)
        am = self._bn(self.first_m[1], ym, training, cache)
        ar = self._bn(self.first_r[1], yr, training, cache)
        return np.concatenate([am, ar], axis=2)                           # Concatenate()([x_mfcc, x_raw])

    def forward(self, x, training=False, seed=0, step=0, cache=None, drop_offset=0):
        cache = {} if cache is None else cache
        cache['stem'] = self._stem(x, training, cache)
        return self._body(cache['stem'], training, seed, step, cache, drop_offset)

    def _body(self, h, training, seed, step, cache, drop_offset):
        dt = self.dtype
        B = h.shape[0]
        for i, blk in enumerate(self.blocks):
            c = {'x': h}
            if 'short' in blk:
                xs = h[:, ::blk['stride'], :]
                Ws = self._p(blk['short'][0]).reshape(blk['cin'], blk['nf'])
                c['xs'], c['Ws'] = xs, Ws
                res = self._bn(blk['short'][1], L.pw_fwd(xs, Ws), training, cache, relu=False)
            else:
                res = h
            w1 = self._p(blk['dw1']).reshape(3, blk['cin'])
            z1 = L.dwconv_fwd(h, w1, 1, (1, 1))
            W1 = self._p(blk['pw1']).reshape(blk['cin'], blk['nf'])
            a1 = self._bn(blk['bn1'], L.pw_fwd(z1, W1), training, cache)
            w2 = self._p(blk['dw2']).reshape(3, blk['nf'])
            z2 = L.dwconv_fwd(a1, w2, 1, (1, 1))
            W2 = self._p(blk['pw2']).reshape(blk['nf'], blk['nf'])
            a2 = self._bn(blk['bn2'], L.pw_fwd(z2, W2), training, cache)
            pooled, arg = maxpool3_same_fwd(a2, blk['stride'])
            c.update(w1=w1, z1=z1, W1=W1, a1=a1, w2=w2, z2=z2, W2=W2, arg=arg)
            cache['blk%d' % i] = c
            h = pooled + res
        feat = h.mean(axis=1)
        if training:
            m = L.dropout_mask(L.dropout_key(seed, step, 1), B * self.C, self.drop_keep,
                               drop_offset * self.C).reshape(B, self.C)
            fd = feat * m / dt(self.drop_keep)
        else:
            m, fd = None, feat
        Wd, bd = self._p('dense_1/kernel'), self._p('dense_1/bias')
        p = L.softmax(fd @ Wd + bd, axis=1)
        cache['tail'] = (m, fd, Wd, p)
        return p

    def loss_and_grads(self, x, y_onehot, seed=0, step=0, drop_offset=0, loss_scale_B=None, relu_masks=None,
                       pool_args=None):
        dt = self.dtype
        cache = {'relu_masks': relu_masks}
        p = self.forward(x, training=True, seed=seed, step=step, cache=cache, drop_offset=drop_offset)
        y_onehot = np.asarray(y_onehot, dtype=dt)
        loss, per, dp = L.cce_fwd_bwd(p, y_onehot)                        # model.py:1657
        B = p.shape[0]
        if loss_scale_B is not None:
            dp = dp * dt(B) / dt(loss_scale_B)
        grads = OrderedDict()
        m, fd, Wd, p = cache['tail']
        dl = L.softmax_bwd(dp, p, axis=1)
        grads['dense_1/kernel'] = fd.T @ dl
        grads['dense_1/bias'] = dl.sum(axis=0)
        dfeat = (dl @ Wd.T) * m / dt(self.drop_keep)
        dh = np.repeat(dfeat[:, None, :], self.T, axis=1) / dt(self.T)
        for i in reversed(range(len(self.blocks))):
            blk, c = self.blocks[i], cache['blk%d' % i]
            arg = c['arg'] if pool_args is None or i not in pool_args else pool_args[i]
            da2 = maxpool3_same_bwd(dh, arg, blk['stride'], blk['Lin'])
            dy2 = self._bn_bwd(blk['bn2'], da2, cache, grads)
            dz2, dW2 = L.pw_bwd(dy2, c['z2'], c['W2'])
            grads[blk['pw2']] = dW2.reshape(1, blk['nf'], blk['nf'])
            da1, dw2 = L.dwconv_bwd(dz2, c['a1'], c['w2'], 1, (1, 1))
            grads[blk['dw2']] = dw2.reshape(1, 3, blk['nf'], 1)
            dy1 = self._bn_bwd(blk['bn1'], da1, cache, grads)
            dz1, dW1 = L.pw_bwd(dy1, c['z1'], c['W1'])
            grads[blk['pw1']] = dW1.reshape(1, blk['cin'], blk['nf'])
            dx, dw1 = L.dwconv_bwd(dz1, c['x'], c['w1'], 1, (1, 1))
            grads[blk['dw1']] = dw1.reshape(1, 3, blk['cin'], 1)
            if 'short' in blk:
                dys = self._bn_bwd(blk['short'][1], dh, cache, grads, relu=False)
                dxs, dWs = L.pw_bwd(dys, c['xs'], c['Ws'])
                grads[blk['short'][0]] = dWs.reshape(1, blk['cin'], blk['nf'])
                dx = dx.copy()
                dx[:, ::blk['stride'], :] += dxs
            else:
                dx = dx + dh
            dh = dx
        cols_m, cols_r = cache['conv1_cols']
        for (name, bidx), cols, sl in ((self.first_m, cols_m, slice(0, self.Cm)),
                                       (self.first_r, cols_r, slice(self.Cm, self.Cm + self.Cr))):
            dy = self._bn_bwd(bidx, dh[:, :, sl], cache, grads)
            W0 = self._p(name)
            B2, Lo, Co = dy.shape
            grads[name] = (cols.T @ dy.reshape(B2 * Lo, Co)).reshape(W0.shape)
        for k in self.l2_names:
            grads[k] = grads[k] + dt(2.0 * L.L2_COEF) * self._p(k)
        return loss, p, OrderedDict((k, grads[k]) for k in self.params), cache
