"""Inference API of the reference (``inference/__init__.py:1-8``): same class names, same registry keys.
The dotted paths in ``task_inference_mapping`` resolve both as ``some_amd.inference.*`` and, through the
top-level ``inference`` shim package of this repository, exactly as the reference spells them."""
from . import base_infer, me_infer, me_quant_infer

BaseInference = base_infer.BaseInference
MIDIExtractionInference = me_infer.MIDIExtractionInference
QuantizedMIDIExtractionInference = me_quant_infer.QuantizedMIDIExtractionInference

# training task class (config key ``task_cls``) -> dotted path of the inference class that serves its checkpoints
task_inference_mapping = {
    f'training.{task}': f'inference.{cls.__name__}'
    for task, cls in (('MIDIExtractionTask', MIDIExtractionInference),
                      ('QuantizedMIDIExtractionTask', QuantizedMIDIExtractionInference))
}

__all__ = ['BaseInference', 'MIDIExtractionInference', 'QuantizedMIDIExtractionInference', 'task_inference_mapping']
