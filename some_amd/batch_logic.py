"""Host-side note <-> word alignment of the DiffSinger-dataset path (reference batch_infer.py:37-134, 172-219).

Pure Python float / ``round(x, 6)`` / string arithmetic that decides the text written into
``transcriptions.csv``; it has to reproduce the reference's strings exactly, so it stays in Python with the
same rounding points.  Fixtures: tests/golden/batch_infer_fns.json and
tests/golden/batch_csv_*.csv, produced by the reference's own batch_infer.py."""
from typing import Dict, List, Sequence, Tuple

from .utils.notes import midi_to_note


def calc_seq(note_midi: float, note_rest: bool) -> str:
    """batch_infer.py:37-46: note name + signed cents (e.g. ``C4+12``, ``A#3-7``), ``rest`` for rests."""
    if note_rest:
        # the reference still evaluates the name first; rests may carry any midi value (0 included)
        return 'rest'
    nearest = round(note_midi, 0)
    cent = int(round(note_midi - nearest, 2) * 100)
    suffix = f'+{cent}' if cent > 0 else ('' if cent == 0 else str(cent))
    return f'{midi_to_note(nearest, unicode=False)}{suffix}'


def notes_from_segments(offsets: Sequence[float], segments: Sequence[Dict]) -> List[dict]:
    """batch_infer.py:55-81: flatten per-chunk note arrays into absolute-time note dicts, rounding to 6
    decimals at the reference's rounding points and clamping start times to be monotonic."""
    notes: List[dict] = []
    for offset, seg in zip(offsets, segments):
        offset = round(offset, 6)
        midis, durs, rests = seg['note_midi'].tolist(), seg['note_dur'].tolist(), seg['note_rest'].tolist()
        assert len(midis) == len(durs) == len(rests)
        elapsed = 0
        for midi, dur, rest in zip(midis, durs, rests):
            dur = round(dur, 6)
            elapsed = round(elapsed, 6)
            note = {
                'start_time': round(offset + elapsed, 6),
                'end_time': round(offset + elapsed + dur, 6),
                'note_seq': calc_seq(midi, rest),
            }
            if notes and note['start_time'] < notes[-1]['end_time']:
                note['start_time'] = notes[-1]['end_time']
            note['note_dur'] = round(note['end_time'] - note['start_time'], 6)
            notes.append(note)
            elapsed += dur
    return notes


def get_word_durs(ph_durs: Sequence[float], ph_nums: Sequence[int]) -> List[Tuple[float, float]]:
    """batch_infer.py:84-94: (start, end) of each word from phoneme durations and phonemes-per-word."""
    words, first, t = [], 0, 0
    for n in ph_nums:
        dur = round(sum(ph_durs[first:first + n]), 6)
        words.append((round(t, 6), round(t + dur, 6)))
        first += n
        t += dur
    return words


def midi_align(notes: List[dict], words: Sequence[Tuple[float, float]], tolerance: float = 0.05) -> List[dict]:
    """batch_infer.py:97-110: snap note edges that fall within ``tolerance`` of a word boundary onto it (later
    boundaries win, as in the reference's loop), drop notes that end up with non-positive duration."""
    bounds = [w[0] for w in words] + [words[-1][1]]
    kept = []
    for note in notes:
        for edge in bounds:
            if edge - tolerance <= note['start_time'] <= edge + tolerance:
                note['start_time'] = edge
            if edge - tolerance <= note['end_time'] <= edge + tolerance:
                note['end_time'] = edge
        note['note_dur'] = round(note['end_time'] - note['start_time'], 6)
        if note['note_dur'] > 0:
            kept.append(note)
    return kept


def get_all_overlap_midis(interval, notes):
    """batch_infer.py:113-122."""
    lo, hi = interval
    return [n for n in notes
            if lo < n['start_time'] < hi or lo < n['end_time'] < hi or (n['start_time'] <= lo and hi <= n['end_time'])]


def get_max_overlap_midi(interval, notes):
    """batch_infer.py:125-134: name of the note with the largest overlap (first one on ties), else ``rest``."""
    best, best_overlap = 'rest', 0
    for n in notes:
        overlap = max(0, min(interval[1], n['end_time']) - max(interval[0], n['start_time']))
        if overlap > best_overlap:
            best, best_overlap = n['note_seq'], overlap
    return best


def align_row(notes: List[dict], ph_dur_field: str, ph_num_field: str, round_midi: bool) -> Tuple[str, str]:
    """batch_infer.py:172-219 for one CSV row: returns the ``note_seq`` and ``note_dur`` column strings."""
    ph_dur = [round(float(x), 6) for x in ph_dur_field.split(' ')]
    ph_num = [int(x) for x in ph_num_field.split(' ')]
    words = get_word_durs(ph_dur, ph_num)
    notes = midi_align(notes, words)
    seq_out, dur_out = [], []
    for start, end in words:
        word_dur = round(end - start, 6)
        if round_midi:
            seq_out.append(get_max_overlap_midi((start, end), notes))
            dur_out.append(word_dur)
            continue
        seqs, durs = [], []
        for n in get_all_overlap_midis((start, end), notes):
            seqs.append(n['note_seq'])
            if n['start_time'] <= start:
                durs.append(round(min(end, n['end_time']) - start, 6))
            elif n['end_time'] >= end:
                durs.append(round(end - max(start, n['start_time']), 6))
            else:
                durs.append(round(n['note_dur'], 6))
        if not seqs:
            seqs.append('rest')
            durs.append(word_dur)
        if round(sum(durs), 6) < word_dur:
            seqs.append('rest')
            durs.append(word_dur - round(sum(durs), 6))
        seq_out.extend(seqs)
        dur_out.extend(durs)
    assert len(seq_out) == len(dur_out)
    return ' '.join(str(x) for x in seq_out), ' '.join(str(round(x, 6)) for x in dur_out)


def align_job(offsets, segments, ph_dur_field: str, ph_num_field: str, round_midi: bool) -> Tuple[str, str]:
    """One CSV row, start to finish (notes_from_segments + align_row): the unit of work batch_infer.py hands to its
    alignment worker processes so that this pure-Python arithmetic runs beside the GPU instead of in front of it."""
    return align_row(notes_from_segments(offsets, segments), ph_dur_field, ph_num_field, round_midi)
