"""Checkpoint -> inference object (what reference infer.py:20-31, batch_infer.py:21-34 and webui.py:24-38 each spell out):
``config.yaml`` beside the checkpoint names the training task, the registry names the inference class for it."""
import importlib
import pathlib
from typing import Tuple

import yaml


def resolve_inference_class(task_cls: str):
    import inference
    try:
        dotted = inference.task_inference_mapping[task_cls]
    except KeyError:
        raise KeyError(f"no inference class registered for task '{task_cls}' "
                       f"(known: {sorted(inference.task_inference_mapping)})") from None
    module_name, _, class_name = dotted.rpartition('.')
    cls = getattr(importlib.import_module(module_name), class_name)
    if not (isinstance(cls, type) and issubclass(cls, inference.BaseInference)):
        raise AssertionError(f'Inference class {cls} is not a subclass of {inference.BaseInference}.')
    return cls


def load_inference(model_path, device=None, verbose: bool = True) -> Tuple[object, dict]:
    from utils.config_utils import print_config
    model_path = pathlib.Path(model_path)
    config = yaml.safe_load(model_path.with_name('config.yaml').read_text(encoding='utf8'))
    if verbose:
        print_config(config)
    return resolve_inference_class(config['task_cls'])(config=config, model_path=model_path, device=device), config
