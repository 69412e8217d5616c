"""The trainer loop's side files: scalar logs and checkpoint retention.

* ``ScalarLog`` - what ``TensorBoardLogger(save_dir=work_dir, name='lightning_logs', version='lastest')`` + ``logger.log_metrics`` give the
  reference (train.py:83-87, training/base_task.py:254-260, 311-316): scalars ``training/<loss>``, ``training/lr``, ``validation/<loss>``,
  ``metrics/<name>`` against the global step.  Written twice: a TensorBoard event file (the TFRecord framing + the ``Event`` /
  ``Summary.Value.simple_value`` protobuf fields encoded by hand - tensorboard is not on this image, its readers take the file) and
  ``scalars.csv`` (step, tag, value, wall_time) for everything else.
* ``CheckpointKeeper`` - ``DsModelCheckpoint`` (utils/training_utils.py:182-256): keep the newest ``num_ckpt_keep`` checkpoints; one that
  falls out of the window stays as a PERMANENT checkpoint when ``permanent_ckpt_start > 0``, ``permanent_ckpt_interval > 9``,
  ``step >= start`` and ``(step - start) % interval == 0`` (configs/base.yaml:63-66)."""
import os
import pathlib
import re
import socket
import struct
import time
from typing import Dict, List, Optional

_CRC_TABLE = None


def crc32c(data: bytes) -> int:
    """CRC-32C (Castagnoli, reflected polynomial 0x82F63B78) - the checksum of the TFRecord framing."""
    global _CRC_TABLE
    if _CRC_TABLE is None:
        tab = []
        for i in range(256):
            c = i
            for _ in range(8):
                c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
            tab.append(c)
        _CRC_TABLE = tab
    c = 0xFFFFFFFF
    for b in data:
        c = _CRC_TABLE[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def _masked(crc: int) -> int:
    return ((((crc >> 15) | (crc << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF


def _varint(n: int) -> bytes:
    out = bytearray()
    while True:
        b = n & 0x7F
        n >>= 7
        if n:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _field_bytes(number: int, payload: bytes) -> bytes:
    return _varint(number << 3 | 2) + _varint(len(payload)) + payload


def encode_event(wall_time: float, step: int, scalars: Optional[Dict[str, float]] = None, file_version: Optional[str] = None) -> bytes:
    """tensorflow.Event: 1 wall_time (double), 2 step (int64), 3 file_version (string) | 5 summary { repeated 1 value { 1 tag, 2 simple_value (float) } }."""
    ev = _varint(1 << 3 | 1) + struct.pack('<d', wall_time) + _varint(2 << 3 | 0) + _varint(step)
    if file_version is not None:
        ev += _field_bytes(3, file_version.encode())
    if scalars:
        summary = b''.join(_field_bytes(1, _field_bytes(1, tag.encode()) + _varint(2 << 3 | 5) + struct.pack('<f', float(v))) for tag, v in scalars.items())
        ev += _field_bytes(5, summary)
    return ev


def frame_record(payload: bytes) -> bytes:
    """TFRecord: uint64 length | masked crc32c(length) | payload | masked crc32c(payload)."""
    head = struct.pack('<Q', len(payload))
    return head + struct.pack('<I', _masked(crc32c(head))) + payload + struct.pack('<I', _masked(crc32c(payload)))


def read_records(path) -> List[bytes]:
    """Payloads of a TFRecord file, checksums verified (the tests' reader)."""
    raw, out, pos = pathlib.Path(path).read_bytes(), [], 0
    while pos < len(raw):
        head = raw[pos:pos + 8]
        (n,) = struct.unpack('<Q', head)
        assert struct.unpack('<I', raw[pos + 8:pos + 12])[0] == _masked(crc32c(head)), 'length checksum'
        payload = raw[pos + 12:pos + 12 + n]
        assert struct.unpack('<I', raw[pos + 12 + n:pos + 16 + n])[0] == _masked(crc32c(payload)), 'payload checksum'
        out.append(payload)
        pos += 16 + n
    return out


class ScalarLog:
    def __init__(self, work_dir, name: str = 'lightning_logs', version: str = 'lastest'):     # ('lastest': the reference's spelling, train.py:86)
        self.dir = pathlib.Path(work_dir) / name / version
        self.dir.mkdir(parents=True, exist_ok=True)
        now = time.time()
        self.event_path = self.dir / f'events.out.tfevents.{int(now)}.{socket.gethostname()}.{os.getpid()}.0'
        self._ev = open(self.event_path, 'ab')
        self._ev.write(frame_record(encode_event(now, 0, file_version='brain.Event:2')))
        new = not (self.dir / 'scalars.csv').exists()
        self._csv = open(self.dir / 'scalars.csv', 'a', encoding='utf8')
        if new:
            self._csv.write('step,tag,value,wall_time\n')

    def log_metrics(self, metrics: Dict[str, float], step: int):
        now = time.time()
        scalars = {k: float(v) for k, v in metrics.items()}
        self._ev.write(frame_record(encode_event(now, int(step), scalars)))
        for k, v in scalars.items():
            self._csv.write(f'{int(step)},{k},{v!r},{now:.3f}\n')
        self._ev.flush()
        self._csv.flush()

    def close(self):
        self._ev.close()
        self._csv.close()


class CheckpointKeeper:
    def __init__(self, work_dir, num_ckpt_keep: int = 5, permanent_ckpt_start: Optional[int] = 0, permanent_ckpt_interval: Optional[int] = 0):
        self.work = pathlib.Path(work_dir)
        self.keep = int(num_ckpt_keep)
        self.start, self.interval = int(permanent_ckpt_start or 0), int(permanent_ckpt_interval or 0)
        self.enable_permanent = self.start > 0 and self.interval > 9                     # utils/training_utils.py:194
        # restart: the newest `keep` files are the window whatever their step (permanence is decided when one LEAVES the window, as
        # _remove_checkpoint does, utils/training_utils.py:243-256); older files are left alone, as Lightning leaves them.
        # keep < 0 (save_top_k = -1): keep everything; keep = 0 (save_top_k = 0): save nothing - see wants()
        found = self.existing(self.work)
        self.window: List[pathlib.Path] = found[-self.keep:] if self.keep > 0 else (found if self.keep < 0 else [])

    def wants(self, step: int) -> bool:
        """Should a checkpoint be written at ``step``?  Lightning's ``save_top_k = 0`` writes none; a permanent step is kept regardless here
        (the reference would never have written it - keeping it costs nothing and loses nothing)."""
        return self.keep != 0 or self.is_permanent(step)

    @staticmethod
    def step_of(path) -> int:
        m = re.search(r'steps_(\d+)', pathlib.Path(path).stem)
        return int(m.group(1)) if m else -1

    @classmethod
    def existing(cls, work_dir) -> List[pathlib.Path]:
        """model_ckpt_steps_*.ckpt of the experiment directory, oldest first; the last one is what a restart resumes from
        (get_latest_checkpoint_path, utils/training_utils.py:259-276)."""
        return sorted((p for p in pathlib.Path(work_dir).glob('model_ckpt_steps_*.ckpt') if cls.step_of(p) >= 0), key=cls.step_of)

    def is_permanent(self, step: int) -> bool:
        return self.enable_permanent and step >= self.start and (step - self.start) % self.interval == 0

    def path_for(self, step: int) -> pathlib.Path:
        return self.work / f'model_ckpt_steps_{step}.ckpt'

    def saved(self, path) -> List[str]:
        """Register a checkpoint that has just been written; remove what falls out of the window (unless permanent).  Returns log lines."""
        path, lines = pathlib.Path(path), []
        if path not in self.window:
            self.window.append(path)
        while self.keep >= 0 and len(self.window) > self.keep:
            if self.keep == 0 and self.is_permanent(self.step_of(self.window[0])):
                lines.append(f'Checkpoint {self.window.pop(0).name} is now permanent.')
                continue
            old = self.window.pop(0)
            if self.is_permanent(self.step_of(old)):
                lines.append(f'Checkpoint {old.name} is now permanent.')
            else:
                old.unlink(missing_ok=True)
                lines.append(f'Removed checkpoint {old.name}.')
        return lines
