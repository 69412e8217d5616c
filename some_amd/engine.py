"""Host-side owner of one libsome_amd handle: config marshalling, weight upload, workspace reuse and the
three device stages (log-mel front end, conformer forward, note decode) on PyTorch-owned HBM buffers.

PyTorch is used only as plumbing here - device allocation (caching allocator), streams and, for the
multi-GPU path, ``torch.distributed`` (RCCL) broadcast of the packed weight arena.  All compute happens in the
HIP kernels behind the C ABI (include/some_amd.h).
"""
import ctypes as C
from typing import Dict, List, Mapping, Optional, Sequence

import numpy as np
import torch

from . import _lib


def make_some_config(config: dict) -> _lib.SomeConfig:
    """Marshal the reference's config.yaml keys (SURVEY.md section 5 'Config / flags') into the C struct."""
    a = config.get('midi_extractor_args') or {}
    c = _lib.SomeConfig()
    c.lay = int(a.get('lay', 0))
    c.dim = int(a.get('dim', 512))
    c.heads = int(a.get('attention_heads', 8))
    c.head_dim = int(a.get('attention_heads_dim', 64))
    c.kernel_size = int(a.get('kernel_size', 31))
    c.indim = int(config.get('units_dim', 80))
    c.outdim = int(config.get('midi_num_bins', 128))
    c.sample_rate = int(config.get('audio_sample_rate', 44100))
    c.hop_size = int(config.get('hop_size', 512))
    c.win_size = int(config.get('win_size', 2048))
    c.fmin = float(config.get('fmin', 40))
    fmax = config.get('fmax', 8000)
    c.fmax = float(fmax) if fmax is not None else 0.0
    c.midi_min = float(config.get('midi_min', 0))
    c.midi_max = float(config.get('midi_max', 127))
    c.midi_deviation = float(config.get('midi_prob_deviation', 1.0))
    c.rest_threshold = float(config.get('rest_threshold', 0.1))
    c.precision = resolve_precision(config)
    return c


PRECISIONS = {'f32': _lib.PRECISION_F32, 'f16x3': _lib.PRECISION_F16X3,
              'f16x3_fast': _lib.PRECISION_F16X3_FAST}     # opt-in: the attention product P V with two terms instead of three (include/some_amd.h)
DEFAULT_PRECISION = 'f16x3'


def resolve_precision(config: dict) -> int:
    """GEMM arithmetic: config key ``some_amd_precision`` > env ``SOME_AMD_PRECISION`` > default.
    'f32' = exact fp32 MFMA; 'f16x3' = fp32-equivalent 3-term split on the f16 matrix pipe; 'f16x3_fast' = f16x3 with
    the attention product P V on two terms (attention output to 2^-12 relative instead of 2^-21): an opt-in speed mode, never the default."""
    import os
    name = config.get('some_amd_precision') or os.environ.get('SOME_AMD_PRECISION') or DEFAULT_PRECISION
    if name not in PRECISIONS:
        raise ValueError(f"unknown some_amd precision '{name}' (choose from {sorted(PRECISIONS)})")
    return PRECISIONS[name]


MAX_FORWARD_FRAMES = 262143      # some_forward's limit (csrc/api.hip)


class ClipBatch:
    """Packed var-len batch descriptor: clip b owns frames [frame_offsets[b], frame_offsets[b+1])."""

    def __init__(self, frame_counts: Sequence[int], device, sample_counts: Optional[Sequence[int]] = None):
        fc = np.asarray(frame_counts, dtype=np.int64)
        self.B = int(fc.shape[0])
        self.frame_counts = fc
        fo = np.zeros(self.B + 1, dtype=np.int64)
        np.cumsum(fc, out=fo[1:])
        if fo[-1] >= 2 ** 30:
            raise ValueError('batch too large: total frames must be < 2**30')
        self.frame_offsets = fo.astype(np.int32)
        self.total_frames = int(fo[-1])
        self.max_frames = int(fc.max()) if self.B else 0
        self.frame_offsets_dev = torch.from_numpy(self.frame_offsets).to(device)
        self.sample_offsets = None
        self.sample_offsets_dev = None
        if sample_counts is not None:
            so = np.zeros(self.B + 1, dtype=np.int64)
            np.cumsum(np.asarray(sample_counts, dtype=np.int64), out=so[1:])
            self.sample_offsets = so
            self.sample_offsets_dev = torch.from_numpy(so).to(device)

    @classmethod
    def from_sample_counts(cls, sample_counts: Sequence[int], hop: int, device) -> 'ClipBatch':
        sc = np.asarray(sample_counts, dtype=np.int64)
        return cls(1 + sc // hop, device, sample_counts=sc)     # spec.py: T = 1 + L // hop


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


class Engine:
    def __init__(self, config: dict, device='cuda', host_only: bool = False):
        """host_only=True builds a handle for the host-side entry points only (weight packing, mel basis);
        it is what the CPU test-suite uses.  Every device stage requires an AMD GPU."""
        self.lib = _lib.load()
        self.device = torch.device('cpu' if host_only else device)
        self.host_only = host_only
        if not host_only and self.device.type != 'cuda':
            raise RuntimeError(
                f"some_amd runs its hot path in HIP kernels on an AMD GPU; device '{device}' is not supported "
                f"(there is no CPU fallback).")
        self.config = config
        self.c_config = make_some_config(config)
        h = C.c_void_p()
        _lib.check(None, self.lib.some_create(C.byref(self.c_config), C.byref(h)))
        self.handle = h
        self.arena: Optional[torch.Tensor] = None
        self._ws: Optional[torch.Tensor] = None
        self._scratch: Optional[torch.Tensor] = None
        self.hop = self.c_config.hop_size
        self.outdim = self.c_config.outdim
        self.indim = self.c_config.indim

    def __del__(self):
        try:
            if getattr(self, 'handle', None):
                self.lib.some_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

    # ---- weights --------------------------------------------------------------------------------
    @property
    def arena_numel(self) -> int:
        return int(self.lib.some_arena_bytes(self.handle)) // 4

    def pack_state_dict(self, state_dict: Mapping[str, 'torch.Tensor']) -> torch.Tensor:
        """strict=True load (base_infer.py:33) into the flat host arena (fp32 CPU tensor)."""
        n = len(state_dict)
        descs = (_lib.SomeTensorDesc * n)()
        keep = []
        for i, (k, v) in enumerate(state_dict.items()):
            t = v if isinstance(v, torch.Tensor) else torch.from_numpy(np.asarray(v))
            t = t.detach().cpu()
            if t.dtype == torch.int64:
                dt = 1
            else:
                dt = 0
                t = t.to(torch.float32)
            t = t.contiguous()
            keep.append(t)
            name = k.encode('utf8')
            keep.append(name)
            descs[i].name = name
            descs[i].data = t.data_ptr()
            descs[i].dtype = dt
            descs[i].ndim = t.dim()
            if t.dim() > 4:
                raise ValueError(f'unexpected tensor rank for {k}: {t.dim()}')
            for d in range(t.dim()):
                descs[i].shape[d] = t.shape[d]
        arena = torch.empty(self.arena_numel, dtype=torch.float32)
        _lib.check(self.handle, self.lib.some_pack_weights(self.handle, descs, n, C.c_void_p(arena.data_ptr())))
        return arena

    def attach_arena(self, arena_dev: torch.Tensor):
        assert arena_dev.is_cuda and arena_dev.dtype == torch.float32 and arena_dev.is_contiguous()
        _lib.check(self.handle, self.lib.some_attach_arena(self.handle, _ptr(arena_dev), arena_dev.numel() * 4))
        self.arena = arena_dev

    def load_state_dict(self, state_dict, strict: bool = True):
        if not strict:
            raise NotImplementedError('only strict=True loading is supported (as the reference inference path uses)')
        host = self.pack_state_dict(state_dict)
        self.attach_arena(host.to(self.device))

    # ---- buffers --------------------------------------------------------------------------------
    def _workspace(self, nbytes: int) -> torch.Tensor:
        if self._ws is None or self._ws.numel() < nbytes:
            self._ws = None
            self._ws = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        return self._ws

    def _decode_scratch(self, nbytes: int) -> torch.Tensor:
        if self._scratch is None or self._scratch.numel() < nbytes:
            self._scratch = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        return self._scratch

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    # ---- stages ---------------------------------------------------------------------------------
    def logmel(self, audio: torch.Tensor, batch: ClipBatch, reflect: bool = False) -> torch.Tensor:
        """audio: packed fp32 [total_samples] on device -> units [total_frames, n_mels].  reflect=True is the
        deployment front end (torch.stft(center=True) reflect padding, deployment/base_onnx_module.py:68-76)."""
        if reflect and int(np.diff(batch.sample_offsets).min()) <= self.c_config.win_size // 2:
            raise ValueError('reflect padding needs clips longer than win_size / 2 samples')
        assert audio.is_cuda and audio.dtype == torch.float32 and audio.is_contiguous()
        assert batch.sample_offsets_dev is not None and audio.numel() == int(batch.sample_offsets[-1])
        units = torch.empty((batch.total_frames, self.indim), dtype=torch.float32, device=self.device)
        _lib.check(self.handle, self.lib.some_logmel(
            self.handle, _ptr(audio), _ptr(batch.sample_offsets_dev), _ptr(batch.frame_offsets_dev),
            batch.B, batch.max_frames, _lib.PAD_REFLECT if reflect else _lib.PAD_ZERO, _ptr(units), self._stream()))
        return units

    def logmel_shifted(self, audio: torch.Tensor, sample_counts: Sequence[int], n_fft_new: int, win_length_new: int,
                       hop_length_new: int, center: bool = True, rescale: bool = True):
        """MelSpectrogram.forward with a key shift / speed change (modules/rmvpe/spec.py:38-72; the training-data
        augmentation of preprocessing/me_binarizer.py:235-246).  audio: packed fp32 [total_samples] on device.
        Returns (units [total_frames, n_mels], ClipBatch)."""
        assert audio.is_cuda and audio.dtype == torch.float32 and audio.is_contiguous()
        sc = np.asarray(sample_counts, dtype=np.int64)
        assert int(sc.sum()) == audio.numel()
        padded = sc + (win_length_new if center else 0)
        if sc.size and int(padded.min()) < n_fft_new:        # torch.stft raises for the same input
            raise RuntimeError(f'clip of {int(sc.min())} samples is shorter than one frame of n_fft={n_fft_new}')
        batch = ClipBatch(1 + (padded - n_fft_new) // hop_length_new, self.device, sample_counts=sc)
        units = torch.empty((batch.total_frames, self.indim), dtype=torch.float32, device=self.device)
        _lib.check(self.handle, self.lib.some_logmel_shifted(
            self.handle, _ptr(audio), _ptr(batch.sample_offsets_dev), _ptr(batch.frame_offsets_dev), batch.B,
            batch.max_frames, int(n_fft_new), int(win_length_new), int(hop_length_new), int(bool(center)),
            int(bool(rescale)), _ptr(units), self._stream()))
        return units, batch

    # ---- host ingest either side of the silence slicer ------------------------------------------------------
    @staticmethod
    def _sample_format(audio: torch.Tensor) -> int:
        if audio.dtype == torch.int16:
            return _lib.SAMPLE_PCM16
        if audio.dtype == torch.float32:
            return _lib.SAMPLE_F32
        raise TypeError(f'audio must be int16 PCM or float32, got {audio.dtype}')

    def slicer_rms(self, audio: torch.Tensor, sample_counts: Sequence[int], frame_length: int, hop_length: int):
        """get_rms (utils/slicer2.py:5-38) of every packed clip on the device, bit-identical to the numpy reduction.
        audio: packed int16 PCM or fp32 [total_samples] on device.  Returns (rms [sum_b (1 + n_b // hop)] device
        fp32, rms_offsets numpy int64 [B + 1])."""
        assert audio.is_cuda and audio.is_contiguous() and audio.dim() == 1
        sc = np.asarray(sample_counts, dtype=np.int64)
        assert int(sc.sum()) == audio.numel()
        so = np.zeros(sc.shape[0] + 1, dtype=np.int64)
        np.cumsum(sc, out=so[1:])
        counts = 1 + sc // int(hop_length)
        ro = np.zeros_like(so)
        np.cumsum(counts, out=ro[1:])
        offs = torch.from_numpy(np.stack([so, ro])).to(self.device)          # one H2D for both offset tables
        rms = torch.empty((int(ro[-1]),), dtype=torch.float32, device=self.device)
        _lib.check(self.handle, self.lib.some_slicer_rms(
            self.handle, _ptr(audio), self._sample_format(audio), _ptr(offs[0]), _ptr(offs[1]), int(sc.shape[0]),
            int(counts.max()) if sc.shape[0] else 0, int(frame_length), int(hop_length), _ptr(rms), self._stream()))
        return rms, ro

    def pcm_gather(self, audio: torch.Tensor, src_offsets: Sequence[int], lengths: Sequence[int]):
        """Cut spans [src_offsets[b], src_offsets[b] + lengths[b]) out of the packed upload (int16 PCM or fp32) into
        the packed fp32 layout ``logmel`` reads.  Returns (audio_f32 device, ClipBatch)."""
        assert audio.is_cuda and audio.is_contiguous() and audio.dim() == 1
        ln = np.asarray(lengths, dtype=np.int64)
        src = np.asarray(src_offsets, dtype=np.int64)
        assert src.shape == ln.shape
        if ln.size and (src.min() < 0 or (src + ln).max() > audio.numel() or ln.min() < 0):
            raise ValueError('pcm_gather: span outside the uploaded audio')
        batch = ClipBatch.from_sample_counts(ln, self.hop, self.device)
        out = torch.empty((int(batch.sample_offsets[-1]),), dtype=torch.float32, device=self.device)
        src_dev = torch.from_numpy(src).to(self.device)
        _lib.check(self.handle, self.lib.some_pcm_gather(
            self.handle, _ptr(audio), self._sample_format(audio), _ptr(src_dev), _ptr(batch.sample_offsets_dev), batch.B,
            int(ln.max()) if ln.size else 0, _ptr(out), self._stream()))
        return out, batch

    def forward(self, units: torch.Tensor, batch: ClipBatch, mask: Optional[torch.Tensor] = None,
                head_mode: int = _lib.HEAD_LOGITS):
        """units [total_frames, indim] -> (midi [total_frames, outdim], bound [total_frames])."""
        assert units.is_cuda and units.dtype == torch.float32 and units.is_contiguous()
        assert units.shape == (batch.total_frames, self.indim), (units.shape, batch.total_frames)
        m = batch.total_frames
        if m > MAX_FORWARD_FRAMES:
            raise ValueError(f'{m} frames in one forward call: the library takes at most {MAX_FORWARD_FRAMES} (32-bit byte offsets into '
                             f'the [frames, 2048] hidden activations; about 50 minutes of audio) - pack fewer clips per batch, or cut the clip')
        if m + 15 * batch.B >= 1048450:
            raise ValueError(f'{batch.B} clips / {m} frames in one forward call: the attention operands (every clip padded to a multiple of 16 rows) '
                             f'must stay below 1 048 450 rows - pack fewer clips per batch')
        midi = torch.empty((m, self.outdim), dtype=torch.float32, device=self.device)
        bound = torch.empty((m,), dtype=torch.float32, device=self.device)
        mask_u8 = None
        if mask is not None:
            mask_u8 = mask.reshape(-1).to(device=self.device, dtype=torch.uint8).contiguous()
            assert mask_u8.numel() == m
        nbytes = int(self.lib.some_workspace_bytes(self.handle, m, batch.B))
        ws = self._workspace(nbytes)
        _lib.check(self.handle, self.lib.some_forward(
            self.handle, _ptr(units), _ptr(batch.frame_offsets_dev), batch.B, m, batch.max_frames,
            _ptr(mask_u8), head_mode, _ptr(midi), _ptr(bound), _ptr(ws), ws.numel(), self._stream()))
        return midi, bound

    def decode(self, probs: torch.Tensor, bounds: torch.Tensor, batch: ClipBatch, quantized: bool,
               mask: Optional[torch.Tensor] = None, debug: bool = False) -> Dict[str, torch.Tensor]:
        """Device decode; outputs are padded per clip (rows frame_offsets[b] .. + n_notes[b])."""
        m = batch.total_frames
        assert probs.is_cuda and probs.dtype == torch.float32 and probs.is_contiguous() and probs.shape == (m, self.outdim)
        assert bounds.is_cuda and bounds.dtype == torch.float32 and bounds.is_contiguous() and bounds.numel() == m
        dev = self.device
        out = {
            'note_midi': torch.empty(m, dtype=torch.float32, device=dev),
            'note_dur': torch.empty(m, dtype=torch.int64, device=dev),
            'note_rest': torch.empty(m, dtype=torch.uint8, device=dev),
            'n_notes': torch.empty(batch.B, dtype=torch.int32, device=dev),
        }
        f2i = val = rest = None
        if debug:
            f2i = out['frame2item'] = torch.empty(m, dtype=torch.int64, device=dev)
            val = out['values'] = torch.empty(m, dtype=torch.float32, device=dev)
            rest = out['rest'] = torch.empty(m, dtype=torch.uint8, device=dev)
        mask_u8 = None
        if mask is not None:
            mask_u8 = mask.reshape(-1).to(device=dev, dtype=torch.uint8).contiguous()
        sc = self._decode_scratch(int(self.lib.some_decode_scratch_bytes(self.handle, m)))
        _lib.check(self.handle, self.lib.some_decode(
            self.handle, _ptr(probs), _ptr(bounds), _ptr(mask_u8), _ptr(batch.frame_offsets_dev), batch.B, m,
            1 if quantized else 0, _ptr(out['note_midi']), _ptr(out['note_dur']), _ptr(out['note_rest']),
            _ptr(out['n_notes']), _ptr(f2i), _ptr(val), _ptr(rest), _ptr(sc), sc.numel(), self._stream()))
        return out

    # ---- graph replay of a whole step ------------------------------------------------------------
    def graph_runner(self, audio: torch.Tensor, batch: ClipBatch, head_mode: int = _lib.HEAD_LOGITS, quantized: bool = False,
                     reflect: bool = False) -> 'GraphStep':
        """log-mel -> forward -> decode for ONE batch shape captured into a hipGraph (every entry point only enqueues on the caller's
        stream, include/some_amd.h): ``runner(new_audio)`` copies the samples into the captured input buffer and replays ~220 launches
        with one hipGraphLaunch.  For callers that see the same (clip lengths) again and again - fixed-length chunks, the latency leg
        of bench.py, a service with a chunk-length grid (the reference's loop being replaced: inference/base_infer.py:46-53)."""
        return GraphStep(self, audio, batch, head_mode, quantized, reflect)

    # ---- measurement ----------------------------------------------------------------------------
    def profile_enable(self, on: bool):
        _lib.check(self.handle, self.lib.some_profile_enable(self.handle, 1 if on else 0))

    def profile_collect(self) -> List[dict]:
        stats = (_lib.SomeKernelStat * 64)()
        n = C.c_int32(0)
        _lib.check(self.handle, self.lib.some_profile_collect(self.handle, stats, 64, C.byref(n)))
        return [dict(name=stats[i].name.decode(), launches=int(stats[i].launches), total_ms=float(stats[i].total_ms),
                     flops=float(stats[i].flops), bytes=float(stats[i].bytes)) for i in range(n.value)]

    def mel_filterbank(self) -> np.ndarray:
        out = np.empty((self.indim, 1 + self.c_config.win_size // 2), dtype=np.float32)
        _lib.check(self.handle, self.lib.some_mel_filterbank(self.handle, C.c_void_p(out.ctypes.data)))
        return out


class GraphStep:
    """One captured step of an Engine for a fixed batch shape (see ``Engine.graph_runner``).  Owns its workspace, decode scratch, input and
    output buffers (the engine may grow or replace its own afterwards); outputs are overwritten by every replay.  Results are bit-identical to
    the eager calls (same kernels, same launch order per stream; tests/test_gpu_parity.py)."""

    def __init__(self, eng: Engine, audio: torch.Tensor, batch: ClipBatch, head_mode: int, quantized: bool, reflect: bool):
        assert audio.is_cuda and audio.dtype == torch.float32 and audio.numel() == int(batch.sample_offsets[-1])
        self.eng, self.batch = eng, batch
        self.audio = audio.clone()                    # the captured input buffer
        saved = (eng._ws, eng._scratch)
        eng._ws = eng._scratch = None                 # buffers of its own, allocated by the warm-up below and kept alive here

        def step():
            units = eng.logmel(self.audio, batch, reflect=reflect)
            probs, bounds = eng.forward(units, batch, head_mode=head_mode)
            out = eng.decode(probs, bounds, batch, quantized=quantized)
            out['probs'], out['bounds'] = probs, bounds
            return out
        try:
            side = torch.cuda.Stream(eng.device)
            side.wait_stream(torch.cuda.current_stream(eng.device))
            with torch.cuda.stream(side):             # warm-up on the capture stream: workspace, helper stream, per-device tables
                step()
            side.synchronize()
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph, stream=side):
                self.out = step()
            self._keep = (eng._ws, eng._scratch)
        finally:
            eng._ws, eng._scratch = saved

    def __call__(self, audio: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
        if audio is not None and audio.data_ptr() != self.audio.data_ptr():
            self.audio.copy_(audio, non_blocking=True)
        self.graph.replay()
        return self.out
