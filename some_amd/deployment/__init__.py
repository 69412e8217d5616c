from .me_onnx_module import MIDIExtractionONNXModule, QuantizedMIDIExtractionONNXModule, MelSpectrogram_ONNX  # noqa: F401
