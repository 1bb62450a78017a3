"""Drop-in module: the reference script imports `model`; the implementation is speech_recognition_amd.model."""
from speech_recognition_amd.model import (overlapping_time_slice_stack, prepare_model_settings, relu6,  # noqa: F401
                                          speech_model)
