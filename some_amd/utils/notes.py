"""``librosa.midi_to_note(midi, unicode=False)`` (call site batch_infer.py:45) restated from librosa 0.9's
published behaviour for the default key (C major, sharps): round to the nearest integer, name = pitch class +
octave number ``int(n / 12) - 1``."""
import numpy as np

_NAMES = ['C', 'C#', 'D', 'D#', 'E', 'F', 'F#', 'G', 'G#', 'A', 'A#', 'B']


def midi_to_note(midi, unicode: bool = False) -> str:
    n = int(np.round(midi))
    name = _NAMES[n % 12]
    if unicode:
        name = name.replace('#', '♯')
    return f'{name}{int(n / 12) - 1:0d}'
