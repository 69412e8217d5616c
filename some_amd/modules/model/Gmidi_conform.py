"""``midi_conforms`` - the model operator of the reference (modules/model/Gmidi_conform.py:22-40), same
constructor and ``forward`` signature, with the body replaced by one call into the HIP library.

It is an ``nn.Module`` so that ``build_object_from_class_name(config['model_cls'], nn.Module, config=config)
.eval().to(device)`` followed by ``load_state_dict(state_dict, strict=True)`` (inference/base_infer.py:24-33)
works unchanged.  It holds no ``nn.Parameter``: weights live in the library's packed device arena.
"""
from typing import Optional

import torch
from torch import nn

from ... import _lib
from ...engine import ClipBatch, Engine


class midi_conforms(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.config = config
        self._engine: Optional[Engine] = None
        self._device = torch.device('cpu')
        self._host_arena = None

    # ---- nn.Module plumbing ------------------------------------------------------------------------
    def _apply(self, fn, recurse=True):
        # .to(device) / .cuda(): find out where we are being sent, then (re)attach the arena there
        probe = fn(torch.empty(0))
        if probe.device.type == 'cuda':      # dtype-only casts (.float(), .eval()) leave the device alone
            self._device = probe.device
            self._ensure_engine()
        return super()._apply(fn, recurse)

    def _ensure_engine(self) -> Engine:
        if self._engine is None or self._engine.device != self._device:
            self._engine = Engine(self.config, device=self._device)     # raises on non-GPU devices
            if self._host_arena is not None:
                self._engine.attach_arena(self._host_arena.to(self._device))
        return self._engine

    @property
    def engine(self) -> Engine:
        return self._ensure_engine()

    def load_state_dict(self, state_dict, strict: bool = True, assign: bool = False):
        """Same contract as nn.Module.load_state_dict(strict=True): missing / unexpected keys or wrong shapes
        raise RuntimeError (the library reports them the way PyTorch words them)."""
        if not strict:
            raise NotImplementedError('strict=False is not supported')
        host = Engine(self.config, host_only=True) if self._device.type != 'cuda' else self._ensure_engine()
        try:
            self._host_arena = host.pack_state_dict(state_dict)
        except _lib.SomeError as e:
            raise RuntimeError(f'Error(s) in loading state_dict for {self.__class__.__name__}: {e}') from e
        if self._device.type == 'cuda':
            self._ensure_engine().attach_arena(self._host_arena.to(self._device))
        return torch.nn.modules.module._IncompatibleKeys([], [])

    def load_packed_arena(self, arena_dev: torch.Tensor):
        """Attach an already packed device arena (what the sharded path receives from the rank-0 broadcast)."""
        self._host_arena = None
        self._device = arena_dev.device
        self._ensure_engine().attach_arena(arena_dev)

    # ---- the operator ------------------------------------------------------------------------------
    @torch.no_grad()
    def forward(self, x, f0=None, mask=None, softmax=False, sig=False):
        """x [B,T,units_dim] fp32, f0 ignored (as in the reference: Gconform.py:119-140 never reads it),
        mask [B,T] bool or None -> (midi [B,T,outdim], bound [B,T])."""
        eng = self._ensure_engine()
        if x.dim() != 3:
            raise ValueError(f'expected x of shape [B, T, {eng.indim}], got {tuple(x.shape)}')
        b, t, _ = x.shape
        batch = ClipBatch([t] * b, eng.device)
        units = x.to(device=eng.device, dtype=torch.float32).reshape(b * t, -1).contiguous()
        if sig and softmax:
            # the reference would apply sigmoid and then softmax (Gmidi_conform.py:33-37); no call site does
            raise NotImplementedError('sig=True together with softmax=True is not supported')
        mode = _lib.HEAD_SIGMOID if sig else (_lib.HEAD_SOFTMAX if softmax else _lib.HEAD_LOGITS)
        midi, bound = eng.forward(units, batch, mask=mask, head_mode=mode)
        return midi.view(b, t, -1), bound.view(b, t)
