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
