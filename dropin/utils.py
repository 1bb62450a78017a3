"""Drop-in module: the reference script imports `utils`; the implementation is speech_recognition_amd.utils."""
from speech_recognition_amd.utils import *  # noqa: F401,F403
from speech_recognition_amd.utils import (center_crop, data_gen, smooth_categorical_crossentropy,  # noqa: F401,E402
                                          tf_roll)
