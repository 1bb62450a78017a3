"""Drop-in module: the reference script imports `callbacks`; the implementation is speech_recognition_amd.callbacks."""
from speech_recognition_amd.callbacks import *  # noqa: F401,F403
from speech_recognition_amd.callbacks import ConfusionMatrixCallback, log_loss  # noqa: F401,E402
