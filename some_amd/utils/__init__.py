"""Host-side helpers mirroring the parts of the reference's ``utils`` package that the inference path uses."""
import importlib

# model_cls strings found in the reference's configs -> our operator (configs/*.yaml `model_cls`)
_MODEL_REGISTRY = {
    'modules.model.Gmidi_conform.midi_conforms': 'some_amd.modules.model.Gmidi_conform.midi_conforms',
}


def build_object_from_class_name(cls_str, parent_cls, *args, **kwargs):
    """Reference utils/__init__.py:221-230: resolve a dotted class path, assert its base, construct it.
    The reference's ``model_cls`` value is transparently mapped to the HIP-backed operator."""
    cls_str = _MODEL_REGISTRY.get(cls_str, cls_str)
    pkg, cls_name = cls_str.rsplit('.', 1)
    cls_type = getattr(importlib.import_module(pkg), cls_name)
    if parent_cls is not None:
        assert issubclass(cls_type, parent_cls), f'| {cls_type} is not subclass of {parent_cls}.'
    return cls_type(*args, **kwargs)
