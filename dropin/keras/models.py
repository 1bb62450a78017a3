"""keras.models.load_model for the .npz state-dicts written by Model.save (make_submission.py:64-71)."""
import numpy as np


def load_model(filepath, custom_objects=None, compile=True):
    from speech_recognition_amd.model import speech_model
    with np.load(filepath) as z:
        name = str(z['__model_name__']) if '__model_name__' in z.files else 'conv_1d_time_sliced_with_attention'
        num_classes = int(z['__num_classes__']) if '__num_classes__' in z.files else 12
        input_size = int(z['__input_size__']) if '__input_size__' in z.files else 16000
        kw = {}
        if '__spectrogram_length__' in z.files and int(z['__spectrogram_length__']) > 0:
            kw = dict(spectrogram_length=int(z['__spectrogram_length__']), num_log_mel_features=int(z['__num_features__']))
    model = speech_model(name, input_size, num_classes=num_classes, **kw)
    model.load_weights(filepath)
    return model


from speech_recognition_amd.keras_api import Model  # noqa: E402,F401
