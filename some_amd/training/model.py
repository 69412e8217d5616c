"""Train-mode twin of ``midi_conforms`` (modules/model/Gmidi_conform.py:30-40 over modules/conform/Gconform.py) on
the HIP training operators.

Parameters live in ONE flat fp32 buffer (``FlatParams``): every state-dict tensor is a view of it, gradients
accumulate into views of one flat gradient buffer, so the optimiser is a single fused AdamW launch and data-parallel
training is a single all-reduce of the flat gradient.  Keys and shapes are the reference's ``state_dict`` ones, so
checkpoints move both ways (``load_state_dict`` / ``state_dict``)."""
from collections import OrderedDict
from typing import Dict, Optional

import numpy as np
import torch

from ..engine import ClipBatch
from ..synth import state_dict_shapes
from .ops import TrainOps

_BUFFER_LEAVES = ('running_mean', 'running_var', 'num_batches_tracked')


class FlatParams:
    def __init__(self, config: dict, device):
        self.shapes = state_dict_shapes(config)
        self.param_names = [k for k in self.shapes if k.rsplit('.', 1)[-1] not in _BUFFER_LEAVES]
        self.buffer_names = [k for k in self.shapes if k.rsplit('.', 1)[-1] in _BUFFER_LEAVES]
        self.offsets: Dict[str, int] = {}
        pos = 0
        for k in self.param_names:
            self.offsets[k] = pos
            pos += (int(np.prod(self.shapes[k])) + 63) // 64 * 64          # 256-byte aligned views
        self.numel = pos
        self.flat = torch.zeros(pos, dtype=torch.float32, device=device)
        self.grad = torch.zeros(pos, dtype=torch.float32, device=device)
        self.views: 'OrderedDict[str, torch.Tensor]' = OrderedDict()
        for k in self.param_names:
            n, off = int(np.prod(self.shapes[k])), self.offsets[k]
            p = self.flat[off:off + n].view(self.shapes[k])
            p.requires_grad_(True)
            p.grad = self.grad[off:off + n].view(self.shapes[k])            # autograd accumulates in place
            self.views[k] = p
        self.buffers: 'OrderedDict[str, torch.Tensor]' = OrderedDict()
        for k in self.buffer_names:
            leaf = k.rsplit('.', 1)[-1]
            if leaf == 'num_batches_tracked':
                self.buffers[k] = torch.zeros((), dtype=torch.int64, device=device)
            else:
                self.buffers[k] = (torch.ones if leaf == 'running_var' else torch.zeros)(self.shapes[k], dtype=torch.float32, device=device)

    def __getitem__(self, key: str) -> torch.Tensor:
        return self.views[key] if key in self.views else self.buffers[key]

    def zero_grad(self):
        self.grad.zero_()

    def init_like_torch(self, seed: int):
        """PyTorch's default initialisers for the reference's layers (nn.Linear / nn.Conv1d: U(-1/sqrt(fan_in), ...) for
        weight and bias; LayerNorm / BatchNorm: ones / zeros)."""
        g = torch.Generator(device='cpu').manual_seed(seed)
        with torch.no_grad():
            for k, p in self.views.items():
                leaf = k.rsplit('.', 1)[-1]
                if '.norm' in k:
                    p.fill_(1.0 if leaf == 'weight' else 0.0)
                    continue
                wkey = k[:-len(leaf)] + 'weight'
                fan_in = int(np.prod(self.shapes[wkey][1:]))
                bound = 1.0 / np.sqrt(fan_in)
                p.copy_(((torch.rand(p.shape, generator=g) * 2 - 1) * bound).to(p.device))

    def load_state_dict(self, sd, strict: bool = True):
        sd = {(k[6:] if k.startswith('model.model.') else k): v for k, v in sd.items()}      # Lightning prefix
        missing = [k for k in self.shapes if k not in sd]
        unexpected = [k for k in sd if k not in self.shapes]
        if strict and (missing or unexpected):
            raise RuntimeError(f'Error(s) in loading state_dict: Missing key(s): {missing}; Unexpected key(s): {unexpected}')
        with torch.no_grad():
            for k in self.shapes:
                if k in sd:
                    v = torch.as_tensor(np.asarray(sd[k]) if not torch.is_tensor(sd[k]) else sd[k])
                    if tuple(v.shape) != tuple(self.shapes[k]):
                        raise RuntimeError(f'size mismatch for {k}: {tuple(v.shape)} vs {tuple(self.shapes[k])}')
                    self[k].copy_(v.to(self[k].device))

    def state_dict(self) -> 'OrderedDict[str, torch.Tensor]':
        return OrderedDict((k, self[k].detach().clone()) for k in self.shapes)


class TrainableMidiConforms:
    """forward(units [M, 80], batch, mask) -> (midi logits [M, outdim], bounds [M]) with ``sig=False`` semantics
    (me_task.py:97): raw midi logits, sigmoid-ed bounds.  ``training`` toggles dropout; BatchNorm always uses batch
    statistics here (train mode) - evaluation goes through the inference engine."""

    def __init__(self, config: dict, ops: TrainOps, seed: int = 114514):
        self.config, self.ops = config, ops
        a = config['midi_extractor_args']
        if a['dim'] != 512 or a['attention_heads'] != 8 or a['attention_heads_dim'] != 64 or a['kernel_size'] != 31:
            raise ValueError('the HIP kernels are built for dim 512, 8 x 64 heads, kernel 31')
        self.lay = a['lay']
        self.drop = {'conv': a.get('conv_drop', 0.1), 'ffn_latent': a.get('ffn_latent_drop', 0.1), 'ffn_out': a.get('ffn_out_drop', 0.1),
                     'attention': a.get('attention_drop', 0.1)}
        self.params = FlatParams(config, ops.device)
        self.params.init_like_torch(seed)
        self.training = True
        self._seed = seed
        self._calls = 0
        self._shadow_list = None
        self._block_cache = {}
        self._tracked = []                          # num_batches_tracked buffers of the blocks run in this pass

    # ---- dropout: one independent (seed, counter) stream per call site and step; fused into the neighbouring pass --------
    def _drop(self, kind: str):
        p = self.drop[kind] if self.training else 0.0
        if p <= 0.0:
            return 0.0, 0
        self._calls += 1
        return p, (self._seed * 1000003 + self._calls) * 4294967311 % (1 << 62)

    # ---- blocks (Gconform.py) ------------------------------------------------------------------------------------------
    def _shadow_weights(self):
        """The matrices the fused sub-blocks read as 16-bit images (ops.shadow16), in a fixed order - refreshed together once per pass."""
        o, P = self.ops, self.params
        if not (o.ffn16 and o._hi_mode in (1, 2)):
            return []
        if self._shadow_list is None:
            ws = []
            for pre in [f'model.cf_lay.{i}.{b}' for i in range(self.lay) for b in ('att1', 'att2')] + ['model.att1', 'model.att2']:
                ws += [P[pre + '.ffn1.ln1.weight'], P[pre + '.ffn1.ln2.weight'], P[pre + '.ffn2.ln1.weight'], P[pre + '.ffn2.ln2.weight'],
                       P[pre + '.att.to_out.0.weight'], P[pre + '.conv.pointwise_conv1.weight'], P[pre + '.conv.pointwise_conv2.weight']]
                wqkv = o.joined(P[pre + '.att.to_q.weight'], P[pre + '.att.to_kv.weight'])[0]
                if wqkv is not None:
                    ws.append(wqkv)
            self._shadow_list = ws
        return self._shadow_list

    def _block_params(self, pre: str):
        """The parameter / buffer tensors of block ``pre`` in call order, looked up once (the views never move): ~60 formatted dictionary
        lookups per block and pass otherwise."""
        hit = self._block_cache.get(pre)
        if hit is None:
            P = self.params
            a, c = pre + '.att', pre + '.conv'
            ffn = lambda i, f: (P[f'{pre}.norm{i}.weight'], P[f'{pre}.norm{i}.bias'], P[f'{pre}{f}.ln1.weight'], P[f'{pre}{f}.ln1.bias'],   # noqa: E731
                                P[f'{pre}{f}.ln2.weight'], P[f'{pre}{f}.ln2.bias'])
            hit = self._block_cache[pre] = {
                'ffn1': ffn(1, '.ffn1'), 'ffn2': ffn(4, '.ffn2'),
                'att': (P[f'{pre}.norm2.weight'], P[f'{pre}.norm2.bias'], P[a + '.to_q.weight'], P[a + '.to_kv.weight'], P[a + '.to_out.0.weight'],
                        P[a + '.to_out.0.bias']),
                'conv': (P[f'{pre}.norm3.weight'], P[f'{pre}.norm3.bias'], P[c + '.pointwise_conv1.weight'], P[c + '.pointwise_conv1.bias'],
                         P[c + '.depthwise_conv.weight'], P[c + '.depthwise_conv.bias'], P[c + '.norm.weight'], P[c + '.norm.bias'],
                         P[c + '.norm.running_mean'], P[c + '.norm.running_var'], P[c + '.pointwise_conv2.weight'], P[c + '.pointwise_conv2.bias']),
                'tracked': P[c + '.norm.num_batches_tracked'],
                'ln5': (P[f'{pre}.norm5.weight'], P[f'{pre}.norm5.bias']),
            }
        return hit

    def _ffn_block(self, x, params):
        """x = ffn(norm_i(x)) * 0.5 + x (Gconform.py:57,60), the FFN's output dropout included - one fused operator in mixed precision."""
        latent, out = self._drop('ffn_latent'), self._drop('ffn_out')                # the call order of the unfused composition
        return self.ops.ffn_block(x, *params, 0.5, latent[0], latent[1], out[0], out[1])

    def _block(self, x, pre: str, batch):
        """conform_blocke.forward (Gconform.py:56-63)."""
        o, bp = self.ops, self._block_params(pre)
        x = self._ffn_block(x, bp['ffn1'])
        x = o.attention_block(x, *bp['att'], batch, *self._drop('attention'))
        x = o.conv_block(x, *bp['conv'], batch, *self._drop('conv'))
        self._tracked.append(bp['tracked'])
        x = self._ffn_block(x, bp['ffn2'])
        return o.layernorm(x, *bp['ln5'])

    def forward(self, units: torch.Tensor, batch: ClipBatch, mask: Optional[torch.Tensor] = None):
        """Gmidi_conform.forward (Gconform.py:119-140) + midi_conforms.forward(sig=False)."""
        P, o = self.params, self.ops
        o.weights_version += 1                     # the parameters may have moved since the last pass: 16-bit weight images are re-derived
        o.prepare_shadows(self._shadow_weights())  # ... all of them in one launch (mixed precision only)
        units = units.reshape(-1, units.shape[-1]).contiguous()
        mask_u8 = None if mask is None else mask.reshape(-1).to(torch.uint8).contiguous()
        x = o.linear(units, P['model.inln.weight'], P['model.inln.bias'])
        x1 = o.linear(units, P['model.inln1.weight'], P['model.inln1.bias'])
        if mask_u8 is not None:
            x = o.mask_rows(x, mask_u8)
        for i in range(self.lay):
            pre = f'model.cf_lay.{i}'
            fork = o.fork_point()                      # the bound stream's block runs on lane 1, behind what is enqueued up to HERE
            m = self._block(x, pre + '.att1', batch)
            with o.lane(1, after=fork):
                b = self._block(x1, pre + '.att2', batch)
            gm = o.glu(o.linear(m, P[pre + '.glu1.0.weight'], P[pre + '.glu1.0.bias']))
            gb = o.glu(o.linear(b, P[pre + '.glu2.0.weight'], P[pre + '.glu2.0.bias']))
            x, x1 = o.axpy(1.0, gb, m), o.axpy(1.0, gm, b)                                   # Gcf.forward :82-87
            if mask_u8 is not None:
                x = o.mask_rows(x, mask_u8)
        fork = o.fork_point()
        x = self._block(x, 'model.att1', batch)
        with o.lane(1, after=fork):
            x1 = self._block(x1, 'model.att2', batch)
        with torch.no_grad():
            torch._foreach_add_(self._tracked, 1)          # BatchNorm's num_batches_tracked of every block: one launch
        self._tracked.clear()
        midi = o.linear(x, P['model.outln.weight'], P['model.outln.bias'])
        bound = o.reshape(o.sigmoid(o.linear(x1, P['model.cutheard.weight'], P['model.cutheard.bias'])), -1)
        return midi, bound

    __call__ = forward

    def train(self, mode: bool = True):
        self.training = mode
        return self

    def eval(self):
        return self.train(False)
