from speech_recognition_amd.keras_api import (Callback, History, ModelCheckpoint, ReduceLROnPlateau,  # noqa: F401
                                              TensorBoard)
