"""Drop-in module: the reference script imports `classes`; the implementation is speech_recognition_amd.classes."""
from speech_recognition_amd.classes import *  # noqa: F401,F403
from speech_recognition_amd.classes import get_classes, get_int2label, get_label2int  # noqa: F401,E402
