"""Drop-in module: the reference script imports `audio`; the implementation is speech_recognition_amd.audio."""
from speech_recognition_amd.audio import *  # noqa: F401,F403
from speech_recognition_amd.audio import AudioConverter  # noqa: F401,E402
