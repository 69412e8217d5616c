"""``BaseInference`` - constructor, attributes and method names of reference inference/base_infer.py:13-53."""
import pathlib
from collections import OrderedDict
from typing import Dict, List

import numpy as np
import torch
from torch import nn

from ..utils import build_object_from_class_name


class BaseInference:
    def __init__(self, config: dict, model_path: pathlib.Path, device=None):
        if device is None:
            device = 'cuda' if torch.cuda.is_available() else 'cpu'    # base_infer.py:15-16
        if torch.device(device).type != 'cuda':
            raise RuntimeError(
                "some_amd needs an AMD GPU (PyTorch-ROCm device 'cuda'); got device "
                f"'{device}'. The hot path has no CPU fallback - use the reference implementation on CPU.")
        self.config = config
        self.model_path = model_path
        self.device = device
        self.timestep = self.config['hop_size'] / self.config['audio_sample_rate']
        self.model: torch.nn.Module = self.build_model()

    def build_model(self, local: bool = False) -> nn.Module:
        """base_infer.py:23-35: build ``model_cls``, load ``ckpt['state_dict']`` entries under ``model.``,
        strict.  ``local=True`` loads on THIS process only, with no collective - for rebuilds that a single rank of a
        sharded job decides on its own (the precision fallback in me_infer.py): the other ranks are not in a broadcast."""
        model: nn.Module = build_object_from_class_name(
            self.config['model_cls'], nn.Module, config=self.config
        ).eval().to(self.device)
        prefix_in_ckpt = 'model'
        import torch.distributed as dist
        # (a process group of ONE rank - torch.distributed.run --nproc-per-node 1 - takes the same path: the broadcast is then RCCL's
        # one-rank case, which keeps the collective path exercised on single-GPU boxes)
        sharded = (not local) and dist.is_available() and dist.is_initialized()
        self.loaded_from_cache = False
        if hasattr(model, 'load_packed_arena'):
            # the packed arena is cached next to the checkpoint (some_amd/arena_cache.py) and, with one process per GPU,
            # read + packed by rank 0 only and sent with ONE broadcast (RCCL over xGMI); other ranks never touch the file
            arena = None
            if not sharded or dist.get_rank() == 0:
                arena = self._packed_arena(model.engine, prefix_in_ckpt)
            if sharded:
                dev = torch.empty(model.engine.arena_numel, dtype=torch.float32, device=self.device)
                if dist.get_rank() == 0:
                    dev.copy_(arena)
                dist.broadcast(dev, src=0)
            else:
                dev = arena.to(self.device)
            model.load_packed_arena(dev)
        else:
            model.load_state_dict(self._read_state_dict(prefix_in_ckpt), strict=True)
        if not sharded or dist.get_rank() == 0:
            print(f'| load \'{prefix_in_ckpt}\' from \'{self.model_path}\'.')
        return model

    def _packed_arena(self, engine, prefix_in_ckpt: str) -> torch.Tensor:
        """Host arena of this checkpoint: from the cache file when it matches the checkpoint, else strict load + pack
        (and the cache is written for the next start).  ``some_amd_arena_cache: false`` in the config turns it off."""
        from .. import arena_cache
        use_cache = bool(self.config.get('some_amd_arena_cache', True))
        prec = int(engine.c_config.precision)
        if use_cache:
            hit = arena_cache.load(pathlib.Path(self.model_path), engine.arena_numel, prec, self.config)
            if hit is not None:
                self.loaded_from_cache = True
                return torch.from_numpy(hit)
        arena = engine.pack_state_dict(self._read_state_dict(prefix_in_ckpt))
        if use_cache:
            arena_cache.store(pathlib.Path(self.model_path), arena.numpy(), prec, self.config)
        return arena

    def _read_state_dict(self, prefix_in_ckpt: str):
        state_dict = torch.load(self.model_path, map_location='cpu')['state_dict']
        return OrderedDict({
            k[len(prefix_in_ckpt) + 1:]: v
            for k, v in state_dict.items() if k.startswith(f'{prefix_in_ckpt}.')
        })

    def preprocess(self, waveform: np.ndarray) -> Dict[str, torch.Tensor]:
        raise NotImplementedError()

    def forward_model(self, sample: Dict[str, torch.Tensor]):
        raise NotImplementedError()

    def postprocess(self, results: Dict[str, torch.Tensor]) -> List[Dict[str, np.ndarray]]:
        raise NotImplementedError()

    def infer(self, waveforms: List[np.ndarray]) -> List[Dict[str, np.ndarray]]:
        """Reference semantics (base_infer.py:46-53): one result dict per waveform, each processed on its own.
        Subclasses batch the clips on the device; the per-clip results are identical because every kernel
        treats the clips of a packed batch independently."""
        results = []
        for w in waveforms:
            results.append(self.postprocess(self.forward_model(self.preprocess(w))))
        return results
