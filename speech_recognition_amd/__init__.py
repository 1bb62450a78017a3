"""speech_recognition_amd - MI355X-native hot path of see--/speech_recognition.

Package layout: csrc/ (HIP kernels + C ABI -> libkws_hip.so), _lib.py (ctypes binding),
net.py (device buffers + network programs), and the host-side mirrors of the reference's
Python interface (input_data, utils, model, callbacks, classes, audio).
"""
__version__ = "0.1.0"
