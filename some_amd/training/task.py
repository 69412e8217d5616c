"""Losses, optimiser, schedule and the training step of ``MIDIExtractionTask`` (training/me_task.py:57-111,
training/base_task.py optimiser wiring, lr_scheduler/scheduler.py:42-59, configs/two_head_model.yaml:38-52) on the HIP
training operators.  Data-parallel: one process per GPU, gradients summed with ONE all-reduce of the flat gradient
buffer (RCCL over xGMI on the GPUs, gloo in the CPU tests) and averaged inside the fused AdamW launch."""
import ctypes as C
import math
import os
import time
from typing import Dict, Optional

import torch

from ..engine import ClipBatch, Engine
from .grad_sync import BucketedGradSync
from .model import TrainableMidiConforms
from .ops import Tape, TrainOps


def warmup_lr(step: int, base_lr: float, warmup_steps: int, min_lr: float) -> float:
    """lr_scheduler.scheduler.WarmupLR: linear warm-up to ``base_lr`` over ``warmup_steps``, then base_lr *
    (warmup_steps / step) ** 0.5 floored at ``min_lr`` (step counts optimiser updates from 1)."""
    step = max(step, 1)
    if warmup_steps == 0:
        return max(base_lr * step ** -0.5, min_lr)
    lr = base_lr * warmup_steps ** 0.5 * min(step ** -0.5, step * warmup_steps ** -1.5)
    return min_lr if (lr < min_lr and step > warmup_steps) else lr


class MIDIExtractionTrainer:
    def __init__(self, config: dict, device='cuda', seed: int = 114514, process_group=None):
        self.config = config
        self.engine = Engine(config, device=device)
        self.ops = TrainOps(self.engine)
        # pl_trainer_precision (configs/base.yaml:74 '32-true'; configs/midi_conformer.yaml:35 'bf16'): any 16-bit setting
        # selects mixed precision - f16 operands on the matrix pipe, fp32 everything else, dynamic loss scaling
        prec = str(config.get('pl_trainer_precision', '32-true'))
        self.mixed = bool(config.get('some_amd_mixed_precision', '16' in prec))
        # 'bf16' / 'bf16-mixed' -> bf16 operands (the reference's arithmetic), '16-mixed' -> f16 operands; some_amd_mixed_operand overrides
        self.mixed_operand = str(config.get('some_amd_mixed_operand', 'bf16' if 'bf16' in prec else 'f16'))
        self.ops.set_mixed_precision(self.mixed, self.mixed_operand)
        self.model = TrainableMidiConforms(config, self.ops, seed=seed)
        oa = config.get('optimizer_args', {})
        self.base_lr = oa.get('lr', 1e-4)
        self.betas = (oa.get('beta1', 0.9), oa.get('beta2', 0.98))
        self.weight_decay = oa.get('weight_decay', 0.0)
        self.eps = oa.get('eps', 1e-8)
        sa = config.get('lr_scheduler_args', {})
        self.warmup_steps, self.min_lr = sa.get('warmup_steps', 5000), sa.get('min_lr', 1e-5)
        n = self.model.params.numel
        self.exp_avg = torch.zeros(n, dtype=torch.float32, device=self.ops.device)
        self.exp_avg_sq = torch.zeros(n, dtype=torch.float32, device=self.ops.device)
        self.global_step = 0
        # Dynamic loss scaling (power of two; exact in fp32): the split-f16 GEMMs carry operands as f16 hi + lo, whose
        # absolute floor is 2^-25 - activation gradients of a mean-reduced loss (~1 / (B T N)) sit below it unscaled.
        # The scaled gradient stays in the flat buffer and is unscaled inside the fused AdamW launch; a non-finite
        # gradient halves the scale and skips the update, `growth_interval` clean updates double it.
        # bf16 operands have the fp32 exponent range: no scaling
        scaled = self.ops.gemm_precision == 'f16x3' and self.ops.operand != 'bf16'
        self.loss_scale = float(config.get('some_amd_loss_scale', 2.0 ** 14)) if scaled else 1.0
        self.growth_interval = int(config.get('some_amd_loss_scale_growth_interval', 200))
        self._clean_steps = 0
        self._sumsq = torch.zeros(1, dtype=torch.float64, device=self.ops.device)
        # training_step(sync=False): gradient-norm checks of updates that have been enqueued but not looked at yet - (event, pinned copy of
        # the squared norm, step) - at most ``_max_in_flight`` of them
        self._pending = []
        self._max_in_flight = 2
        self._norm_slots = None               # pinned ring for those copies (allocated at the first asynchronous update)
        # SOME_AMD_TRAIN_TAPE=0 / some_amd_tape: false: forward + backward through torch.autograd (A/B runs; same kernels, same gradients)
        self.use_tape = bool(config.get('some_amd_tape', True)) and os.environ.get('SOME_AMD_TRAIN_TAPE', '1') != '0'
        self.host_enqueue_s = 0.0                 # cumulative host time spent enqueuing training steps (up to the step's one sync)
        self.pg = process_group
        self.world = 1
        if process_group is not None or torch.distributed.is_initialized():
            self.world = torch.distributed.get_world_size(process_group)
            torch.distributed.broadcast(self.model.params.flat, src=0, group=process_group)      # identical replicas
        # gradient all-reduce in buckets, overlapped with the backward pass (grad_sync.py); some_amd_ddp_overlap=False falls back
        # to one all-reduce of the whole flat gradient after backward
        self.grad_sync = None
        if self.world > 1 and config.get('some_amd_ddp_overlap', True):
            P = self.model.params
            order = [(P.views[k], P.offsets[k], (P.views[k].numel() + 63) // 64 * 64) for k in P.param_names]
            self.grad_sync = BucketedGradSync(P.grad, order, process_group, int(config.get('some_amd_ddp_bucket_mb', 32)) << 20,
                                              names=list(P.param_names), static_graph=bool(config.get('some_amd_ddp_static_graph', False)))
        # Gradient sinks: nn.Linear / LayerNorm parameter gradients are written into the flat buffer by the backward kernels themselves
        # (ops.py: no per-parameter copy / accumulation launches).  Under data parallelism each deposit is reported to the bucketed
        # sync through mark().  (Round 2 kept sinks off for world > 1 because the replicas diverged: autograd fires a parameter's
        # post-accumulate hook even when backward returned None for it, so mark() + hook counted every sunk parameter twice and buckets
        # were all-reduced when half of their gradients were in the buffer.  BucketedGradSync now ignores that echo and raises on any
        # other double report - tests/test_train_host.py::test_gradient_sync_mark_stands_in_for_the_hook.)
        if self.grad_sync is not None:
            # a bucket holds gradients written on both lanes (ops.py): its all-reduce is ordered behind the issuing lane's stream only
            self.grad_sync.before_launch = self.ops.sync_other_lane
        if config.get('some_amd_grad_sinks', True):
            self.ops.register_grad_sinks(self.model.params.views.values(), self.grad_sync.mark if self.grad_sync is not None else None)

    # ---- me_task.py:79-111 ------------------------------------------------------------------------------------
    def run_model(self, sample: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        """sample: 'units' [B, T, 80], 'unit2note' [B, T] int, 'probs' [B, T, N], 'bounds' [B, T] (the collater's
        batch, me_task.py:26-52).  Returns the loss dict of run_model(infer=False)."""
        units = sample['units']
        B, T = units.shape[0], units.shape[1]
        batch = ClipBatch([T] * B, self.ops.device)
        mask = sample['unit2note'] > 0
        probs, bounds = self.model(units.reshape(B * T, -1), batch, mask=mask)
        losses = {}
        if self.config.get('use_bound_loss', True):
            losses['bound_loss'] = self.ops.binary_emd(bounds, sample['bounds'].reshape(-1).float(), B, T)
        if self.config.get('use_midi_loss', True):
            losses['midi_loss'] = self.ops.bce_with_logits(probs, sample['probs'].reshape(B * T, -1).float())
        return losses

    def training_step(self, sample, sync: bool = True) -> Dict[str, float]:
        """One optimiser update: forward, losses, backward, gradient all-reduce, AdamW with the WarmupLR rate.  A list of
        batches is one update over ``accumulate_grad_batches`` micro-batches (configs/base.yaml:50, train.py:89): their
        gradients accumulate in the flat buffer, each loss weighted 1 / n as Lightning does.

        ``sync=False`` (honoured when no loss scaling is active: bf16 operands or exact-f32 GEMMs): the update is enqueued WITHOUT reading the
        gradient norm back - the clip factor is computed on the device (some_train_adamw_clip: the same double arithmetic, bit-identical
        parameters) - so the host goes on to enqueue the next step while this one's tail runs; the returned losses are device tensors,
        ``grad_norm`` is None, and a non-finite gradient (which leaves the parameters untouched) raises FloatingPointError from a LATER
        call or from ``flush()``.  At most two updates are in flight.  After that error the trainer's state is INVALID (later updates were
        already applied with the step counter advanced past the skipped one): training_step / checkpoint refuse until load_checkpoint."""
        self._refuse_if_failed()
        t_begin = time.perf_counter()
        self.ops.pin_stream()
        try:
            return self._training_step(sample, t_begin, sync)
        finally:
            self.ops.unpin_stream()

    def _refuse_if_failed(self):
        if getattr(self, 'failed_step', None) is not None:
            raise RuntimeError(f'the trainer state is invalid since the non-finite gradient of update {self.failed_step} '
                               f'(asynchronous updates ran on past it): reload a checkpoint with load_checkpoint()')

    def flush(self):
        """Wait for the updates enqueued with ``sync=False`` and raise if one of them saw a non-finite gradient."""
        self._check_pending(block=True)

    def _check_pending(self, block: bool, keep: int = 0):
        while len(self._pending) > keep:
            ev, host, step = self._pending[0]
            if not block and not ev.query():
                return
            ev.synchronize()
            self._pending.pop(0)
            v = float(host[0])
            if not (v == v and v != float('inf')):
                # The bad update left parameters and moments untouched on the device, but up to _max_in_flight later updates have been
                # enqueued on top with the step counter (AdamW bias correction, WarmupLR) already advanced past it.  The trainer is NOT
                # state-equivalent to a synchronous run from here on: drop the run-ahead bookkeeping, mark the trainer, and make every
                # later training_step / checkpoint refuse until the caller reloads a checkpoint (load_checkpoint clears the mark).
                self._pending.clear()
                self.failed_step = step
                raise FloatingPointError(f'non-finite gradient in update {step} (its parameters were left untouched; {self.global_step - step} later '
                                         f'update(s) were already enqueued - reload a checkpoint before continuing, the trainer state is invalid)')

    def _training_step(self, sample, t_begin, sync: bool = True) -> Dict[str, float]:
        P = self.model.params
        P.zero_grad()
        self.model.train()
        micro = sample if isinstance(sample, (list, tuple)) else [sample]
        scale = self.loss_scale
        losses, total = {}, 0.0
        for i, mb in enumerate(micro):
            weight = scale / len(micro)
            if self.use_tape:
                # the trainer's own tape instead of torch.autograd (ops.Tape: same operator bodies, a third of the host time per step)
                tape = self.ops.tape = Tape(self.ops)
                try:
                    with torch.no_grad():
                        part = self.run_model(mb)
                    part_total = sum(part.values())
                    if self.grad_sync is not None and i == len(micro) - 1:
                        self.grad_sync.arm()           # buckets go out as this backward pass completes them
                    tape.backward([(v, weight) for v in part.values()])
                finally:
                    self.ops.tape = None
            else:
                part = self.run_model(mb)
                part_total = sum(part.values())
                if self.grad_sync is not None and i == len(micro) - 1:
                    self.grad_sync.arm()               # buckets go out as this backward pass completes them
                (part_total * weight if weight != 1.0 else part_total).backward()
            total = total + part_total.detach() / len(micro)
            for k, v in part.items():
                losses[k] = losses.get(k, 0.0) + v.detach() / len(micro)
        if self.grad_sync is not None:
            self.grad_sync.finish()
        elif self.world > 1:
            torch.distributed.all_reduce(P.grad, op=torch.distributed.ReduceOp.SUM, group=self.pg)
        # global gradient norm on the device (one double comes back: the step's only host synchronisation); it serves
        # Lightning's gradient_clip_val = clip_grad_norm (configs/base.yaml:49, train.py:88) and the overflow check
        sc = self.ops.scratch(1, 1)
        self.ops.check(self.ops.lib.some_train_sumsq(self.ops.h, P.grad.data_ptr(), P.numel, self._sumsq.data_ptr(),
                                                     sc.data_ptr(), sc.numel(), self.ops.stream()))
        clip = self.config.get('clip_grad_norm', None)
        if not sync and scale == 1.0:
            self._check_pending(block=False)
            self._check_pending(block=True, keep=self._max_in_flight - 1)       # bounded run-ahead
            lr = warmup_lr(self.global_step + 1, self.base_lr, self.warmup_steps, self.min_lr)
            self.global_step += 1
            p = lambda t: t.data_ptr()  # noqa: E731
            self.ops.check(self.ops.lib.some_train_adamw_clip(self.ops.h, p(P.flat), p(P.grad), p(self.exp_avg), p(self.exp_avg_sq), P.numel, lr,
                                                              self.betas[0], self.betas[1], self.eps, self.weight_decay, self.global_step,
                                                              p(self._sumsq), float(clip or 0.0), float(self.world * scale), self.ops.stream()))
            self.ops.weights_version += 1
            if self._norm_slots is None:
                self._norm_slots = torch.empty(2 * self._max_in_flight, dtype=torch.float64).pin_memory()
            k = self.global_step % self._norm_slots.numel()
            host = self._norm_slots[k:k + 1]
            host.copy_(self._sumsq, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            self._pending.append((ev, host, self.global_step))
            self.host_enqueue_s += time.perf_counter() - t_begin
            out = dict(losses)
            out.update(total_loss=total, lr=lr, grad_scale=scale, skipped=False, grad_norm=None)
            return out
        self._check_pending(block=True)
        self.host_enqueue_s += time.perf_counter() - t_begin      # host time to enqueue the step, up to its one synchronisation
        sumsq = float(self._sumsq.item())
        skipped = False
        grad_norm = float('nan')
        if not (sumsq == sumsq and sumsq != float('inf')):
            if scale == 1.0:
                raise FloatingPointError('non-finite gradient')
            self.loss_scale, self._clean_steps, skipped = scale * 0.5, 0, True
        else:
            grad_norm = math.sqrt(sumsq) / (scale * self.world)       # (sqrt: correctly rounded, as on the device - adamw_clip_kernel)
            if scale != 1.0:
                self._clean_steps += 1
                if self._clean_steps % self.growth_interval == 0 and self.loss_scale < 2.0 ** 24:
                    self.loss_scale = scale * 2.0
        clip_coef = min(1.0, clip / (grad_norm + 1e-6)) if (clip and not skipped) else 1.0       # torch.nn.utils.clip_grad_norm_
        lr = warmup_lr(self.global_step + 1, self.base_lr, self.warmup_steps, self.min_lr)
        if not skipped:
            self.global_step += 1
            p = lambda t: t.data_ptr()  # noqa: E731
            self.ops.check(self.ops.lib.some_train_adamw(self.ops.h, p(P.flat), p(P.grad), p(self.exp_avg), p(self.exp_avg_sq), P.numel, lr,
                                                         self.betas[0], self.betas[1], self.eps, self.weight_decay, self.global_step,
                                                         clip_coef / (self.world * scale), self.ops.stream()))
            self.ops.weights_version += 1          # the parameters moved: cached 16-bit weight images (ops.shadow16) are stale for ANY caller
        out = dict(losses)
        out['total_loss'] = total
        out['lr'] = lr
        out['grad_scale'] = scale                # P.grad holds scale * (sum over ranks of) the gradient
        out['skipped'] = skipped
        out['grad_norm'] = grad_norm
        return out

    # ---- checkpoint / resume (train.py:98-108: Lightning resumes from the latest checkpoint of the work dir) ------------
    def checkpoint(self) -> Dict[str, object]:
        """Lightning-layout checkpoint: ``state_dict`` with the ``model.`` prefix (what the inference classes read) plus
        everything needed to continue bit-identically: step, flat AdamW moments, loss-scale state, dropout call counter."""
        P = self.model.params
        self._refuse_if_failed()
        self.flush()
        return {
            'state_dict': {'model.' + k: v.cpu() for k, v in P.state_dict().items()},
            'global_step': self.global_step,
            'some_amd_trainer': {
                'exp_avg': self.exp_avg.cpu(), 'exp_avg_sq': self.exp_avg_sq.cpu(), 'loss_scale': self.loss_scale,
                'clean_steps': self._clean_steps, 'dropout_calls': self.model._calls, 'dropout_seed': self.model._seed,
                'param_names': list(P.param_names),
            },
        }

    def load_checkpoint(self, ckpt: Dict[str, object]):
        P = self.model.params
        P.load_state_dict({(k[6:] if k.startswith('model.') else k): v for k, v in ckpt['state_dict'].items()})
        self.ops.weights_version += 1              # (see _training_step)
        self.global_step = int(ckpt.get('global_step', 0))
        self.failed_step = None
        self._pending.clear()
        st = ckpt.get('some_amd_trainer')
        if st is not None:                      # a reference / inference-only checkpoint has no optimiser state: fresh moments
            if list(st['param_names']) != list(P.param_names):
                raise RuntimeError('checkpoint optimiser state does not match this model')
            self.exp_avg.copy_(st['exp_avg'])
            self.exp_avg_sq.copy_(st['exp_avg_sq'])
            self.loss_scale, self._clean_steps = float(st['loss_scale']), int(st['clean_steps'])
            self.model._calls, self.model._seed = int(st['dropout_calls']), int(st['dropout_seed'])

    # ---- me_task.py:113-153 -----------------------------------------------------------------------------------
    @torch.no_grad()
    def sync_eval_engine(self):
        """Pack the current parameters (+ BatchNorm running statistics) into the inference engine: evaluation runs
        the INFERENCE kernels - eval-mode semantics (dropout off, BatchNorm folded from running stats) for free."""
        self.flush()
        self.engine.load_state_dict({k: v.cpu() for k, v in self.model.params.state_dict().items()})

    @torch.no_grad()
    def validation_step(self, sample: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        """_validation_step: eval-mode losses, then sigmoid probabilities -> masked decode -> frame-level MIDIAccuracy
        counts (modules/metrics/midi_acc.py:15-41, tolerance 0.5).  Call ``sync_eval_engine()`` after the last update."""
        from .. import _lib
        units = sample['units']
        B, T = units.shape[0], units.shape[1]
        batch = ClipBatch([T] * B, self.ops.device)
        masks = sample['unit2note'] > 0
        flat_units = units.reshape(B * T, -1).contiguous()
        logits, bounds = self.engine.forward(flat_units, batch, mask=masks, head_mode=_lib.HEAD_LOGITS)
        out: Dict[str, torch.Tensor] = {}
        if self.config.get('use_bound_loss', True):
            out['bound_loss'] = self.ops.binary_emd(bounds, sample['bounds'].reshape(-1).float(), B, T)
        if self.config.get('use_midi_loss', True):
            out['midi_loss'] = self.ops.bce_with_logits(logits, sample['probs'].reshape(B * T, -1).float())
        probs = self.ops.eltwise(_lib.ELT_SIGMOID_FWD, logits)
        dec = self.engine.decode(probs, bounds, batch, quantized=False, mask=masks, debug=True)
        midi_pred = dec['values'].reshape(B, T)
        rest_pred = dec['rest'].reshape(B, T).bool()
        note_midi_gt = sample['note_midi'].float().clone()
        note_midi_gt[sample['note_rest'].bool()] = -torch.inf
        midi_gt = torch.gather(torch.nn.functional.pad(note_midi_gt, [1, 0], value=-torch.inf), 1, sample['unit2note'])
        rest_gt = midi_gt < 0
        close = ~rest_pred & ~rest_gt & (torch.abs(midi_pred - midi_gt) <= 0.5)
        overall = close & (rest_pred == rest_gt) & masks
        out['midi_acc_correct'], out['midi_acc_total'] = overall.sum(), masks.sum()
        out['notes'] = dec['n_notes']
        return out


class QuantizedMIDIExtractionTrainer(MIDIExtractionTrainer):
    """``QuantizedMIDIExtractionTask`` (training/me_quant_task.py:30-78; configs/quant_two_head_model.yaml): the same model with 129 output
    classes (128 = rest), raw logits out of the midi head in training (``softmax=infer``, me_quant_task.py:61), nn.CrossEntropyLoss with
    ignore_index -1 over the frames (me_quant_task.py:42,77) instead of the blurred-target BCE; the bound stream and everything around
    the step (AdamW, WarmupLR, clipping, data parallelism) are the base trainer's."""

    def run_model(self, sample: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        units = sample['units']
        B, T = units.shape[0], units.shape[1]
        batch = ClipBatch([T] * B, self.ops.device)
        mask = sample['unit2note'] > 0
        logits, bounds = self.model(units.reshape(B * T, -1), batch, mask=mask)
        losses = {}
        if self.config.get('use_bound_loss', True):
            losses['bound_loss'] = self.ops.binary_emd(bounds, sample['bounds'].reshape(-1).float(), B, T)
        if self.config.get('use_midi_loss', True):
            losses['midi_loss'] = self.ops.cross_entropy(logits, sample['midi_idx'].reshape(-1), ignore_index=-1)
        return losses

    @torch.no_grad()
    def validation_step(self, sample: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        """_validation_step (me_quant_task.py:81-130): eval-mode losses, then argmax classes -> rest = class 128 -> frame-level
        MIDIAccuracy counts against the per-frame ground truth (tolerance 0.5)."""
        from .. import _lib
        units = sample['units']
        B, T = units.shape[0], units.shape[1]
        batch = ClipBatch([T] * B, self.ops.device)
        masks = sample['unit2note'] > 0
        logits, bounds = self.engine.forward(units.reshape(B * T, -1).contiguous(), batch, mask=masks, head_mode=_lib.HEAD_LOGITS)
        out: Dict[str, torch.Tensor] = {}
        if self.config.get('use_bound_loss', True):
            out['bound_loss'] = self.ops.binary_emd(bounds, sample['bounds'].reshape(-1).float(), B, T)
        if self.config.get('use_midi_loss', True):
            out['midi_loss'] = self.ops.cross_entropy(logits, sample['midi_idx'].reshape(-1), ignore_index=-1)
        cls = logits.reshape(B, T, -1).argmax(dim=-1)
        rest_pred = cls == 128
        midi_pred = cls.float()
        midi_pred[rest_pred] = -torch.inf
        note_midi_gt = sample['note_midi'].float().clone()
        note_midi_gt[sample['note_midi'] == 128] = -torch.inf
        midi_gt = torch.gather(torch.nn.functional.pad(note_midi_gt, [1, 0], value=-torch.inf), 1, sample['unit2note'])
        rest_gt = midi_gt < 0
        close = ~rest_pred & ~rest_gt & (torch.abs(midi_pred - midi_gt) <= 0.5)
        overall = close & (rest_pred == rest_gt) & masks                  # modules/metrics/midi_acc.py:28-32 (a rest frame is never 'close')
        out['midi_acc_correct'], out['midi_acc_total'] = overall.sum(), masks.sum()
        return out


TRAINERS = {'training.MIDIExtractionTask': MIDIExtractionTrainer, 'training.QuantizedMIDIExtractionTask': QuantizedMIDIExtractionTrainer}
