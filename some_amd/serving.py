"""Persistent extraction service: the request handler of the reference's web UI (webui.py:21-66) as a library object,
without Gradio.  The reference keeps one inference instance per checkpoint (webui.py:17, 24-38) and lets up to ten
queue workers call it concurrently (webui.py:104); here concurrent callers are coalesced: ONE dispatcher thread owns
the GPU stream, drains whatever requests arrived while the previous batch was running and sends them through
``infer_files`` as one packed device batch (upload as stored, RMS + chunk cut on the device, silence decisions on the
host).  No request waits for a timer: an idle service starts a lone request at once (B = 1 latency), a busy one batches.

    svc = ExtractionService(work_dir='experiments')
    midi_path, stats = svc.extract_midi('some_model/model.ckpt', 'song.wav', tempo=120)      # webui.infer semantics
    fut = svc.submit('some_model/model.ckpt', pcm_int16)                                     # -> Future of [(offset_s, notes)]
"""
import collections
import pathlib
import queue
import threading
import time
from concurrent.futures import Future
from typing import Dict, Optional, Tuple

import numpy as np
import yaml

MAX_DURATION_S = 20 * 60          # webui.py:43-44


class ExtractionService:
    def __init__(self, work_dir=None, device=None, max_batch_frames: int = 131072, max_models: int = 4):
        self.work_dir = pathlib.Path(work_dir) if work_dir is not None else None
        self.device = device
        self.max_batch_frames = max_batch_frames
        self.max_models = max(1, int(max_models))
        self._instances: 'collections.OrderedDict[str, Tuple[object, dict]]' = collections.OrderedDict()
        self._queue: 'queue.Queue' = queue.Queue()
        self._closed = False
        self.batches_run = 0              # observability: device batches / requests served so far
        self.requests_served = 0
        self._thread = threading.Thread(target=self._dispatch, name='some-amd-dispatch', daemon=True)
        self._thread.start()

    # ---- model cache (webui.py:24-40) -------------------------------------------------------------------
    def resolve_model(self, model_path) -> pathlib.Path:
        """Map a client-supplied model name to a checkpoint file.  With a ``work_dir`` (the served case) only ``*.ckpt``
        files INSIDE it are accepted - the reference's web UI offers exactly ``work_dir.rglob('*.ckpt')`` in a closed
        dropdown (webui.py:82-88, allow_custom_value=False) - so absolute paths, ``..`` and symlinks that leave the
        directory are refused.  Without a ``work_dir`` (library use by trusted code) any existing checkpoint path goes."""
        path = pathlib.Path(model_path)
        if self.work_dir is not None:
            root = self.work_dir.resolve()
            if path.is_absolute():
                raise PermissionError(f'model must be a path relative to the work directory: {model_path}')
            path = (root / path).resolve()
            if not path.is_relative_to(root):
                raise PermissionError(f'model is outside the work directory: {model_path}')
        else:
            path = path.resolve()
        if path.suffix != '.ckpt' or not path.is_file():
            raise FileNotFoundError(f'no such checkpoint: {model_path}')
        return path

    def _instance(self, model_path) -> Tuple[object, dict]:
        """Called on the dispatcher thread only.  Instances are cached per RESOLVED checkpoint path (different spellings
        of one file share an instance) in a small LRU (``max_models``) so clients cannot pile models up in HBM."""
        path = self.resolve_model(model_path)
        key = str(path)
        if key in self._instances:
            self._instances.move_to_end(key)
            return self._instances[key]
        from .inference.loader import resolve_inference_class
        with open(path.with_name('config.yaml'), 'r', encoding='utf8') as f:
            config = yaml.safe_load(f)
        ins = resolve_inference_class(config['task_cls'])(config=config, model_path=path, device=self.device)
        ins.max_batch_frames = self.max_batch_frames
        self._instances[key] = (ins, config)
        while len(self._instances) > self.max_models:
            self._instances.popitem(last=False)
        return self._instances[key]

    # ---- request side -------------------------------------------------------------------------------------
    def submit(self, model_path, samples: np.ndarray) -> Future:
        """samples: one whole mono file, int16 PCM as stored or float32 in [-1, 1].  The future resolves to
        [(chunk offset in seconds, {'note_midi', 'note_dur', 'note_rest'}), ...] (``infer_files`` of one file)."""
        if self._closed:
            raise RuntimeError('ExtractionService is closed')
        if samples.ndim != 1 or samples.dtype not in (np.int16, np.float32):
            raise ValueError('samples must be a mono int16 or float32 array')
        fut: Future = Future()
        self._queue.put((str(model_path), samples, fut))
        return fut

    def extract_midi(self, model_rel_path, input_audio_path, tempo_value, output_midi_path=None):
        """webui.py:21-66 for one uploaded file: returns (midi path or None, statistics / error string)."""
        from .utils.audio import load_pcm
        from .utils.infer_utils import build_midi_file
        if not model_rel_path or not input_audio_path or tempo_value is None:
            return None, 'Error: required inputs not specified.'
        input_audio_path = pathlib.Path(input_audio_path)
        try:
            _, config = self._call_on_dispatcher(lambda: self._instance(model_rel_path))
        except (PermissionError, FileNotFoundError):
            return None, f'Error: unknown model: {model_rel_path}'
        try:
            samples, sr = load_pcm(input_audio_path, sr=config['audio_sample_rate'])
        except Exception:  # noqa: BLE001  (webui.py:48-49: any decode failure is reported, not raised)
            return None, f'Error: unsupported or corrupt file format: {input_audio_path.name}'
        total_duration = samples.shape[0] / sr
        if total_duration > MAX_DURATION_S:
            return None, 'Error: the input audio is too long (>= 20 minutes).'
        start_time = time.time()
        segments = self.submit(model_rel_path, samples).result()
        infer_time = time.time() - start_time
        rtf = infer_time / max(total_duration, 1e-9)
        midi_file = build_midi_file([off for off, _ in segments], [seg for _, seg in segments], tempo=tempo_value)
        out = pathlib.Path(output_midi_path) if output_midi_path is not None else input_audio_path.with_suffix('.mid')
        midi_file.save(out)
        return out, f'Cost {round(infer_time, 2)} s, RTF: {round(rtf, 3)}'

    def _call_on_dispatcher(self, fn):
        fut: Future = Future()
        self._queue.put((None, fn, fut))
        return fut.result()

    def close(self):
        if not self._closed:
            self._closed = True
            self._queue.put(None)
            self._thread.join()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # ---- dispatcher -----------------------------------------------------------------------------------------
    def _dispatch(self):
        from .utils.slicer2 import Slicer
        slicers: Dict[int, Slicer] = {}
        while True:
            item = self._queue.get()
            if item is None:
                return
            batch = [item]
            while True:                                   # everything that queued up while the GPU was busy
                try:
                    nxt = self._queue.get_nowait()
                except queue.Empty:
                    break
                if nxt is None:
                    self._queue.put(None)                 # finish this batch, then stop
                    break
                batch.append(nxt)
            by_model: Dict[str, list] = {}
            for key, payload, fut in batch:
                if key is None:                           # control call (model load)
                    self._resolve(fut, payload)
                    continue
                by_model.setdefault(key, []).append((payload, fut))
            for key, reqs in by_model.items():
                try:
                    ins, config = self._instance(key)
                    sr = config['audio_sample_rate']
                    slicer = slicers.setdefault(sr, Slicer(sr=sr, max_sil_kept=1000))       # webui.py:52
                    results = ins.infer_files([r[0] for r in reqs], slicer)
                    self.batches_run += 1
                    self.requests_served += len(reqs)
                    for (_, fut), res in zip(reqs, results):
                        fut.set_result(res)
                except BaseException as e:  # noqa: BLE001
                    for _, fut in reqs:
                        if not fut.done():
                            fut.set_exception(e)

    @staticmethod
    def _resolve(fut: Future, fn):
        try:
            fut.set_result(fn())
        except BaseException as e:  # noqa: BLE001
            fut.set_exception(e)
