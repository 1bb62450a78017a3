"""Shell of the one librosa entry point the reference uses (create_tta_set.py:1,19:
`from librosa import effects` -> `effects.time_stretch(data, 0.9)`), served by the device kernel."""
from . import effects  # noqa: F401
