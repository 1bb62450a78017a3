"""Oracle: layer-level forward/backward restatements (rows a7-a15).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Channels-last [B, L, C] everywhere (Keras `channels_last`).  TF convolutions are
cross-correlations (no kernel flip).  Every function cites the reference call
site whose TF-1.4 / Keras-2.1.2 op it restates; constants are the ones pinned by
the reference's shipped graph_defs.
"""
import numpy as np

BN_EPS = 1e-3          # batchnorm/add/y               
BN_MOMENTUM = 0.99     # AssignMovingAvg/decay = 0.01  
L2_COEF = 1e-5         # kernel_regularizer=l2(1e-5)   (model.py:37,807,820,829)
KERAS_EPS = 1e-7       # K.epsilon(), loss clip consts (utils.py:103-105)

# ----------------------------------------------------------------------------
# counter-based dropout RNG shared bit-for-bit with the HIP kernels
# ----------------------------------------------------------------------------
_M32 = np.uint64(0xFFFFFFFF)


def fmix32(h):
    """murmur3 finaliser on uint32 (numpy array or int), wrap-around arithmetic."""
    h = np.asarray(h, dtype=np.uint64) & _M32
    h ^= h >> np.uint64(16)
    h = (h * np.uint64(0x85EBCA6B)) & _M32
    h ^= h >> np.uint64(13)
    h = (h * np.uint64(0xC2B2AE35)) & _M32
    h ^= h >> np.uint64(16)
    return h


def dropout_key(seed, step, layer_id):
    """32-bit key of one dropout layer at one step (same derivation in csrc/common.h)."""
    seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    k = int(fmix32((seed & 0xFFFFFFFF) ^ 0x85EBCA6B))
    k = int(fmix32(k ^ (seed >> 32)))
    k = int(fmix32((k + (int(step) & 0xFFFFFFFF) * 0x9E3779B1) & 0xFFFFFFFF))
    k = int(fmix32(k ^ ((int(layer_id) * 0xC2B2AE35) & 0xFFFFFFFF)))
    return k


def dropout_threshold(keep_prob):
    return min(int(keep_prob * 4294967296.0), 0xFFFFFFFF)


def dropout_mask(key, n_total, keep_prob, offset=0):
    """keep[i] = fmix32((offset+i)*0x9E3779B1 + key) < floor(keep_prob * 2^32).

    Keras Dropout(rate) -> tf.nn.dropout(x, keep_prob=1-rate): keep with
    probability keep_prob, scale kept values by 1/keep_prob.  The TF
    random stream itself cannot be reproduced; only the distribution is."""
    idx = (np.arange(n_total, dtype=np.uint64) + np.uint64(offset)) & _M32
    h = fmix32(((idx * np.uint64(0x9E3779B1)) + np.uint64(key)) & _M32)
    return h < np.uint64(dropout_threshold(keep_prob))


# ----------------------------------------------------------------------------
# padding arithmetic
# ----------------------------------------------------------------------------


def same_pad(L, k, stride):
    """TF 'SAME': L' = ceil(L/s); p = max((L'-1)*s + k - L, 0); left = p//2, rest right."""
    Lout = -(-L // stride)
    p = max((Lout - 1) * stride + k - L, 0)
    return Lout, p // 2, p - p // 2


def valid_len(L, k, stride):
    return (L - k) // stride + 1


# ----------------------------------------------------------------------------
# a7: overlapping_time_slice_stack (model.py:67-76) = extract_image_patches SAME
# ----------------------------------------------------------------------------


def frame_same(x, ksize=40, stride=20):
    """[B, L] -> [B, ceil(L/stride), ksize]; F[b,t,j] = x[b, stride*t - pad_l + j] (0 outside)."""
    B, L = x.shape
    Lout, pl, pr = same_pad(L, ksize, stride)
    xp = np.pad(x, [[0, 0], [pl, pr]])
    idx = stride * np.arange(Lout)[:, None] + np.arange(ksize)[None, :]
    return xp[:, idx]


# ----------------------------------------------------------------------------
# a8: Conv1D(k, stride, valid/same, no bias)  (model.py:807, 1450)
# ----------------------------------------------------------------------------


def conv1d_fwd(x, W, stride=1, pad=(0, 0)):
    """x [B,L,Cin], W [k,Cin,Cout] -> [B,Lout,Cout]; y[b,t,o] = sum_{j,c} xpad[b,s*t+j,c] W[j,c,o]."""
    k, Cin, Cout = W.shape
    xp = np.pad(x, [[0, 0], [pad[0], pad[1]], [0, 0]])
    Lout = valid_len(xp.shape[1], k, stride)
    idx = stride * np.arange(Lout)[:, None] + np.arange(k)[None, :]
    cols = xp[:, idx, :]                       # [B, Lout, k, Cin]
    cols2 = cols.reshape(x.shape[0] * Lout, k * Cin)
    y = cols2 @ W.reshape(k * Cin, Cout)
    return y.reshape(x.shape[0], Lout, Cout), cols2


def conv1d_bwd(dy, cols2, W, x_shape, stride=1, pad=(0, 0), need_dx=True):
    k, Cin, Cout = W.shape
    B, Lout, _ = dy.shape
    dy2 = dy.reshape(B * Lout, Cout)
    dW = (cols2.T @ dy2).reshape(k, Cin, Cout)
    dx = None
    if need_dx:
        dcols = (dy2 @ W.reshape(k * Cin, Cout).T).reshape(B, Lout, k, Cin)
        Lp = x_shape[1] + pad[0] + pad[1]
        dxp = np.zeros((B, Lp, Cin), dtype=dy.dtype)
        for j in range(k):
            dxp[:, j:j + stride * Lout:stride, :] += dcols[:, :, j, :]
        dx = dxp[:, pad[0]:pad[0] + x_shape[1], :]
    return dx, dW


# ----------------------------------------------------------------------------
# a9: DepthwiseConv2D((1,3)) on [B,1,L,C]  (model.py:34-44)
# ----------------------------------------------------------------------------


def dwconv_fwd(x, w, stride, pad):
    """x [B,L,C], w [k,C]; y[b,t,c] = sum_j xpad[b, s*t+j, c] w[j,c]."""
    k = w.shape[0]
    xp = np.pad(x, [[0, 0], [pad[0], pad[1]], [0, 0]])
    Lout = valid_len(xp.shape[1], k, stride)
    y = np.zeros((x.shape[0], Lout, x.shape[2]), dtype=x.dtype)
    for j in range(k):
        y += xp[:, j:j + stride * Lout:stride, :] * w[j][None, None, :]
    return y


def dwconv_bwd(dy, x, w, stride, pad):
    k = w.shape[0]
    B, Lout, C = dy.shape
    xp = np.pad(x, [[0, 0], [pad[0], pad[1]], [0, 0]])
    dxp = np.zeros_like(xp)
    dw = np.zeros_like(w)
    for j in range(k):
        sl = slice(j, j + stride * Lout, stride)
        dxp[:, sl, :] += dy * w[j][None, None, :]
        dw[j] = np.sum(dy * xp[:, sl, :], axis=(0, 1))
    dx = dxp[:, pad[0]:pad[0] + x.shape[1], :]
    return dx, dw


# ----------------------------------------------------------------------------
# a10: pointwise Conv1D(num_filter, 1, no bias)  (model.py:48-49)
# ----------------------------------------------------------------------------


def pw_fwd(x, W):
    B, L, Cin = x.shape
    return (x.reshape(B * L, Cin) @ W).reshape(B, L, W.shape[1])


def pw_bwd(dy, x, W):
    B, L, Cin = x.shape
    dy2 = dy.reshape(B * L, -1)
    return (dy2 @ W.T).reshape(B, L, Cin), x.reshape(B * L, Cin).T @ dy2


# ----------------------------------------------------------------------------
# a11: BatchNormalization + relu6  (model.py:46-51, 809-810)
# ----------------------------------------------------------------------------


def bn_train_fwd(y, gamma, beta, eps=BN_EPS):
    """tf.nn.moments over axes [0,1] (biased var) + tf.nn.batch_normalization."""
    mean = y.mean(axis=(0, 1))
    var = ((y - mean) ** 2).mean(axis=(0, 1))
    rstd = 1.0 / np.sqrt(var + y.dtype.type(eps))
    inv = rstd * gamma
    out = y * inv + (beta - mean * inv)
    return out, (mean, var, rstd)


def bn_infer_fwd(y, gamma, beta, mov_mean, mov_var, eps=BN_EPS):
    inv = gamma / np.sqrt(mov_var + y.dtype.type(eps))
    return y * inv + (beta - mov_mean * inv)


def bn_train_bwd(dout, y, gamma, stats):
    mean, var, rstd = stats
    n = y.shape[0] * y.shape[1]
    xhat = (y - mean) * rstd
    dbeta = dout.sum(axis=(0, 1))
    dgamma = (dout * xhat).sum(axis=(0, 1))
    dy = (gamma * rstd) * (dout - dbeta / n - xhat * (dgamma / n))
    return dy, dgamma, dbeta


def bn_moving_update(moving, batch, momentum=BN_MOMENTUM):
    """AssignMovingAvg: m -= (m - batch) * (1 - momentum); variance uses the biased batch var."""
    return moving - (moving - batch) * moving.dtype.type(1.0 - momentum)


def relu6(x):
    """model.py:30-31  K.relu(x, max_value=6) = clip(relu(x), 0, 6)."""
    return np.minimum(np.maximum(x, 0), 6)


def relu6_mask(pre):
    """Gradient mask of relu followed by clip_by_value(0, 6): ReluGrad (pre > 0) times the
    Minimum gradient (value <= 6, inclusive)."""
    return ((pre > 0) & (pre <= 6)).astype(pre.dtype)


def softmax(z, axis=-1):
    z = z - z.max(axis=axis, keepdims=True)
    e = np.exp(z)
    return e / e.sum(axis=axis, keepdims=True)


def softmax_bwd(dp, p, axis=-1):
    return p * (dp - (dp * p).sum(axis=axis, keepdims=True))


# ----------------------------------------------------------------------------
# a13: losses
# ----------------------------------------------------------------------------


def smooth_cce_fwd_bwd(p, y_onehot, label_smoothing=0.1, eps=KERAS_EPS):
    """utils.py:87-108: logits = log(clip(p, eps, 1-eps)); tf.losses.softmax_cross_entropy
    with label smoothing (y*(1-s) + s/C), reduction SUM_BY_NONZERO_WEIGHTS (= batch mean).
    Returns (mean loss, per-sample loss, dL/dp)."""
    B, C = p.shape
    dt = p.dtype.type
    ysm = y_onehot * dt(1.0 - label_smoothing) + dt(label_smoothing / C)
    pc = np.clip(p, dt(eps), dt(1.0 - eps))
    S = pc.sum(axis=1, keepdims=True)
    per = -(ysm * (np.log(pc) - np.log(S))).sum(axis=1)
    dpc = (-ysm / pc + ysm.sum(axis=1, keepdims=True) / S) / dt(B)
    inside = ((p >= dt(eps)) & (p <= dt(1.0 - eps))).astype(p.dtype)
    return per.mean(), per, dpc * inside


def cce_fwd_bwd(p, y_onehot, eps=KERAS_EPS):
    """keras.losses.categorical_crossentropy on softmax output (model.py:1477): Keras
    renormalises p /= sum(p), clips to [eps, 1-eps] and returns -sum(y log p)."""
    B, C = p.shape
    dt = p.dtype.type
    s = p.sum(axis=1, keepdims=True)
    pn = p / s
    pc = np.clip(pn, dt(eps), dt(1.0 - eps))
    per = -(y_onehot * np.log(pc)).sum(axis=1)
    inside = ((pn >= dt(eps)) & (pn <= dt(1.0 - eps))).astype(p.dtype)
    dpn = (-y_onehot / pc) * inside / dt(B)
    dp = dpn / s - (dpn * p).sum(axis=1, keepdims=True) / (s * s)
    return per.mean(), per, dp


def log_loss(y_true, y_pred, eps=1e-12):
    """callbacks.py:6-10."""
    y_pred = np.clip(y_pred, eps, 1.0 - eps)
    return (-(y_true * np.log(y_pred)).sum(axis=1)).mean()


# ----------------------------------------------------------------------------
# a14: optimizers
# ----------------------------------------------------------------------------


def rmsprop_step(p, g, a, lr, rho=0.9, eps=1e-8):
    """Keras 2.1.2 RMSprop: a' = rho a + (1-rho) g^2 ; p' = p - lr g / (sqrt(a') + eps)."""
    dt = p.dtype.type
    a2 = dt(rho) * a + dt(1.0 - rho) * g * g
    p2 = p - dt(lr) * g / (np.sqrt(np.maximum(a2, 0)) + dt(eps))
    return p2, a2


def sgd_momentum_step(p, g, v, lr, momentum=0.9):
    """Keras 2.1.2 SGD (nesterov=False): v' = m v - lr g ; p' = p + v'."""
    dt = p.dtype.type
    v2 = dt(momentum) * v - dt(lr) * g
    return p + v2, v2


# ----------------------------------------------------------------------------
# a17 / a18
# ----------------------------------------------------------------------------


def tta_transform(X, kind):
    """make_submission.py:125-134: 0 identity, 1 np.roll(X,-1500,axis=1), 2 1.2*X,
    3 clip(1.1*X,-1,1), 4 0.9*X."""
    if kind == 0:
        return X
    if kind == 1:
        return np.roll(X, -1500, axis=1)
    if kind == 2:
        return X.dtype.type(1.2) * X
    if kind == 3:
        return np.clip(X.dtype.type(1.1) * X, -1.0, 1.0)
    if kind == 4:
        return X.dtype.type(0.9) * X
    raise ValueError(kind)


def head32to12(p32, all_classes, wanted_classes):
    """freeze_graph_32_classes.py:55-69: [silence, max(unknown-type probs), wanted words in
    all_classes order] -> softmax over those 12 probabilities."""
    mapped = [p32[:, 0]]
    unknown = [p32[:, 1]]
    for i, c in enumerate(all_classes):
        if c in wanted_classes:
            mapped.append(p32[:, i + 2])
        else:
            unknown.append(p32[:, i + 2])
    unk = np.stack(unknown, axis=1).max(axis=1)
    z = np.stack([mapped[0], unk] + mapped[1:], axis=1)
    return softmax(z, axis=1)
