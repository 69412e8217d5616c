"""Inference API of the reference (``inference/__init__.py:1-8``): same class names, same registry keys.
The dotted paths in ``task_inference_mapping`` resolve both as ``some_amd.inference.*`` and, through the
top-level ``inference`` shim package of this repository, exactly as the reference spells them."""
from .base_infer import BaseInference
from .me_infer import MIDIExtractionInference
from .me_quant_infer import QuantizedMIDIExtractionInference

task_inference_mapping = {
    'training.MIDIExtractionTask': 'inference.MIDIExtractionInference',
    'training.QuantizedMIDIExtractionTask': 'inference.QuantizedMIDIExtractionInference',
}

__all__ = ['BaseInference', 'MIDIExtractionInference', 'QuantizedMIDIExtractionInference', 'task_inference_mapping']
