"""Import-compatible shell of the handful of `tensorflow` symbols the reference's scripts touch
(train.py:24-26: GPUOptions, ConfigProto, Session; make_submission.py: sess.run on AudioProcessor
fetches).  No TensorFlow semantics: device work goes through libkws_hip.so."""


class GPUOptions(object):
    def __init__(self, **kw):
        self.__dict__.update(kw)


class ConfigProto(object):
    def __init__(self, **kw):
        self.__dict__.update(kw)


class Session(object):
    """`sess` is passed through the reference opaquely (utils.py:34); `run` evaluates AudioProcessor
    graph outputs (background_clamp_ / spectrogram_ / mfcc_) for one feed_dict."""

    def __init__(self, config=None, graph=None, **_):
        self.config = config

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False

    def run(self, fetches, feed_dict=None):
        if isinstance(fetches, (list, tuple)):
            return [self.run(f, feed_dict) for f in fetches]
        owner = getattr(fetches, 'owner', None)
        if owner is None:
            raise NotImplementedError("Session.run supports AudioProcessor fetches only")
        return owner.run_fetch(fetches, feed_dict or {})

    def close(self):
        pass


def Graph():
    return None
