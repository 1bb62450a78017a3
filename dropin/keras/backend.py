"""keras.backend subset: session bookkeeping (no-ops) and optimizer-variable access."""
import numpy as np

_SESSION = [None]


def set_session(sess):
    _SESSION[0] = sess


def get_session():
    if _SESSION[0] is None:
        import tensorflow as tf
        _SESSION[0] = tf.Session()
    return _SESSION[0]


def set_learning_phase(value):
    pass   # predict() always runs in inference mode, train_on_batch in training mode


def epsilon():
    return 1e-7


def get_value(x):
    return np.float32(getattr(x, 'value', x))


def set_value(x, value):
    x.value = np.float32(value)
