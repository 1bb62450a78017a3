"""Import-compatible shell of the Keras names used by the reference's scripts; the training loop itself
is speech_recognition_amd.keras_api."""
from . import backend, callbacks, models, activations  # noqa: F401
