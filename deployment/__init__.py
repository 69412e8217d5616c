from some_amd.deployment import MIDIExtractionONNXModule, QuantizedMIDIExtractionONNXModule, MelSpectrogram_ONNX  # noqa: F401  (drop-in shim)
