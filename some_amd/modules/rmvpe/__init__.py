from .spec import MelSpectrogram  # noqa: F401
