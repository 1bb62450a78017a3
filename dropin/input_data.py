"""Drop-in module: the reference script imports `input_data`; the implementation is speech_recognition_amd.input_data."""
from speech_recognition_amd.input_data import *  # noqa: F401,F403
from speech_recognition_amd.input_data import (AudioProcessor, load_wav_file, prepare_words_list,  # noqa: F401,E402
                                               save_wav_file, which_set)
