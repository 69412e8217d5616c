"""Minimal Standard MIDI File (type 1) object model + writer standing in for the ``mido`` package, which the
reference uses at utils/infer_utils.py:80-99 (MidiFile / MidiTrack / MetaMessage('set_tempo') /
Message('note_on'|'note_off')) and infer.py:41-42 (``midi_file.save``).  mido is a third-party dependency that is
not vendored in the reference and absent here; the byte layout below follows the SMF 1.0 specification the way
mido's writer emits it (480 ticks per beat default, running status between equal channel status bytes,
``end_of_track`` appended when missing, default velocity 64).  Parity with mido's exact bytes is unpinned by
the reference (no test there)."""
import pathlib
from typing import List


def bpm2tempo(bpm: float) -> int:
    """microseconds per quarter note (mido.bpm2tempo)."""
    return int(round(60 * 1e6 / bpm))


class MetaMessage:
    is_meta = True

    def __init__(self, type, time=0, **kw):
        self.type, self.time, self.kw = type, time, kw
        for k, v in kw.items():
            setattr(self, k, v)

    def bytes(self) -> List[int]:
        if self.type == 'set_tempo':
            t = int(self.tempo)
            return [0xFF, 0x51, 0x03, (t >> 16) & 0xFF, (t >> 8) & 0xFF, t & 0xFF]
        if self.type == 'end_of_track':
            return [0xFF, 0x2F, 0x00]
        raise ValueError(f'unsupported meta message {self.type}')

    def __repr__(self):
        return f'MetaMessage({self.type!r}, time={self.time}, {self.kw})'


class Message:
    is_meta = False
    _STATUS = {'note_off': 0x80, 'note_on': 0x90}

    def __init__(self, type, note=0, velocity=64, time=0, channel=0):
        if type not in self._STATUS:
            raise ValueError(f'unsupported message {type}')
        if not 0 <= int(note) <= 127:
            raise ValueError('data byte must be in range 0..127')
        self.type, self.note, self.velocity, self.time, self.channel = type, int(note), int(velocity), time, int(channel)

    def bytes(self) -> List[int]:
        return [self._STATUS[self.type] | self.channel, self.note, self.velocity]

    def __repr__(self):
        return f'Message({self.type!r}, note={self.note}, velocity={self.velocity}, time={self.time})'


class MidiTrack(list):
    pass


def _vlq(value: int) -> List[int]:
    if value < 0:
        raise ValueError('variable length quantity must be >= 0')
    out = [value & 0x7F]
    value >>= 7
    while value:
        out.append((value & 0x7F) | 0x80)
        value >>= 7
    return out[::-1]


class MidiFile:
    def __init__(self, type=1, ticks_per_beat=480, charset='latin1'):
        self.type, self.ticks_per_beat, self.charset = type, ticks_per_beat, charset
        self.tracks: List[MidiTrack] = []

    def _track_bytes(self, track) -> bytes:
        data = bytearray()
        msgs = list(track)
        if not msgs or msgs[-1].type != 'end_of_track':
            msgs.append(MetaMessage('end_of_track', time=0))
        running = None
        for msg in msgs:
            if int(msg.time) != msg.time:
                raise ValueError('message time must be int in MIDI file')
            data.extend(_vlq(int(msg.time)))
            raw = msg.bytes()
            if msg.is_meta:
                data.extend(raw)
                running = None
            else:
                if raw[0] == running:
                    data.extend(raw[1:])
                else:
                    data.extend(raw)
                running = raw[0]
        return bytes(data)

    def to_bytes(self) -> bytes:
        out = bytearray(b'MThd' + (6).to_bytes(4, 'big') + self.type.to_bytes(2, 'big') +
                        len(self.tracks).to_bytes(2, 'big') + self.ticks_per_beat.to_bytes(2, 'big'))
        for tr in self.tracks:
            body = self._track_bytes(tr)
            out += b'MTrk' + len(body).to_bytes(4, 'big') + body
        return bytes(out)

    def save(self, filename):
        pathlib.Path(filename).write_bytes(self.to_bytes())
