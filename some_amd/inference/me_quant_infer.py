"""``QuantizedMIDIExtractionInference`` (reference inference/me_quant_infer.py:10-38): softmax head over 129
bins (128 = rest), argmax decode."""
from .. import _lib
from .me_infer import MIDIExtractionInference


class QuantizedMIDIExtractionInference(MIDIExtractionInference):
    quantized = True
    head_mode = _lib.HEAD_SOFTMAX          # forward_model passes softmax=True (me_quant_infer.py:13)
