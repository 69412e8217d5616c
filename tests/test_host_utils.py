"""Host-side (CPU) logic of the drop-in layer against fixtures produced by the reference's own code."""
import json

import numpy as np
import pytest

from some_amd import synth
from some_amd.utils.slicer2 import Slicer, get_rms


def _slicer_cases():
    cases = {
        'sil8': synth.synth_clip(3, 20.0, silence_every=4.0),
        'sil5': synth.synth_clip(4, 26.0, silence_every=7.0),
        'nosil': synth.synth_clip(5, 8.0),
        'short': synth.synth_clip(6, 3.0, silence_every=1.0),
    }
    lead = synth.synth_clip(7, 12.0, silence_every=5.0)
    lead[:int(1.7 * 44100)] = 0
    lead[-int(2.2 * 44100):] = 0
    cases['lead_trail'] = lead
    return cases


@pytest.mark.parametrize('name', ['sil8', 'sil5', 'nosil', 'short', 'lead_trail'])
def test_slicer_matches_reference(golden_dir, name):
    want = json.loads((golden_dir / 'slicer.json').read_text())[name]
    w = _slicer_cases()[name]
    chunks = Slicer(sr=44100, max_sil_kept=1000).slice(w)
    got = [[float(c['offset']), int(c['waveform'].shape[0])] for c in chunks]
    assert got == want
    rms_ref = np.load(golden_dir / 'slicer_rms.npz')[name + '.rms']
    if rms_ref.size:
        rms = get_rms(y=w, frame_length=3528, hop_length=882).squeeze(0)
        np.testing.assert_array_equal(rms.astype(np.float32), rms_ref)   # bit-identical RMS


# ---- batch_infer host logic (reference batch_infer.py:37-134) -----------------------------------------
def test_batch_logic_functions_match_reference(golden_dir):
    import copy
    from some_amd import batch_logic as bl
    g = json.loads((golden_dir / 'batch_infer_fns.json').read_text())
    for midi, want in g['calc_seq']:
        assert bl.calc_seq(midi, False) == want
        assert bl.calc_seq(midi, True) == 'rest'
    for c in g['cases']:
        words = bl.get_word_durs(c['ph_dur'], c['ph_num'])
        assert [list(w) for w in words] == c['words']
        for n in c['notes']:
            assert bl.calc_seq(n['_midi'], n['_rest']) == n['note_seq']
        aligned = bl.midi_align(copy.deepcopy(c['notes']), words)
        assert aligned == c['aligned']
        for w, want in zip(words, c['per_word']):
            assert bl.get_max_overlap_midi(w, aligned) == want['max']
            got = [s['note_seq'] + '@' + repr(s['start_time']) for s in bl.get_all_overlap_midis(w, aligned)]
            assert got == want['all']


@pytest.mark.parametrize('stub', ['FakeInference', 'FakeIngestInference'])
@pytest.mark.parametrize('tag,round_midi', [('round', True), ('full', False)])
def test_batch_infer_csv_text_matches_reference(golden_dir, tmp_path, tag, round_midi, stub, monkeypatch):
    """Our batch_infer command with the model stubbed out == the reference's command with the same stub,
    byte for byte (CSV order, skipped missing wav, rounding, rest filling) - both through the reference's generic
    ``infer`` interface and through the whole-file ``infer_files`` entry point the HIP inference classes add."""
    import batch_infer as bi
    import dataset_util
    from some_amd.configs import get_config
    dataset_util.build_dataset(tmp_path)
    monkeypatch.setattr(bi, 'model_init', lambda p: (getattr(dataset_util, stub)(), get_config('midi_conformer')))
    out = tmp_path / f'out_{tag}.csv'
    bi.batch_infer.callback(dataset=str(tmp_path), model=str(tmp_path / 'm.ckpt'), round_midi=round_midi, csv=str(out), overwrite=True)
    assert out.read_bytes() == (golden_dir / f'batch_csv_{tag}.csv').read_bytes()
    with pytest.raises(FileExistsError):
        bi.batch_infer.callback(dataset=str(tmp_path), model=str(tmp_path / 'm.ckpt'), round_midi=round_midi, csv=str(out), overwrite=False)


def test_midi_file_messages_and_bytes(golden_dir):
    from some_amd.utils.infer_utils import build_midi_file
    g = json.loads((golden_dir / 'midi_msgs.json').read_text())
    for c in g.values():
        segs = [{'note_midi': np.array(s['note_midi'], np.float32), 'note_dur': np.array(s['note_dur'], np.float64),
                 'note_rest': np.array(s['note_rest'], bool)} for s in c['segments']]
        mf = build_midi_file(c['offsets'], segs, tempo=c['tempo'])
        got = [[m.type, getattr(m, 'note', -1), int(m.time), int(getattr(m, 'tempo', -1))] for m in mf.tracks[0]]
        assert got == c['messages']
        raw = mf.to_bytes()
        assert raw[:14] == b'MThd' + bytes([0, 0, 0, 6, 0, 1, 0, 1, 0x01, 0xE0])       # type 1, 1 track, 480 tpb
        assert raw[14:18] == b'MTrk' and int.from_bytes(raw[18:22], 'big') == len(raw) - 22
        assert raw[-3:] == bytes([0xFF, 0x2F, 0x00])                                   # end_of_track
        assert raw[22:29] == bytes([0x00, 0xFF, 0x51, 0x03]) + int(round(60e6 / c['tempo'])).to_bytes(3, 'big')


def test_smf_whole_file_known_answer():
    """A two-note file assembled BY HAND from the SMF 1.0 specification (header chunk, track chunk length, variable-length deltas,
    status bytes, set_tempo and end_of_track meta events) - every byte of what ``build_midi_file(...).to_bytes()`` writes, not only the
    header and the tail: note 60 for 0.5 s, a 0.25 s rest (emits nothing, shows up as the next delta), note 64 for 1.0 s at tempo 120
    (480 ticks per beat -> 960 ticks per second).  mido's exact bytes are unpinned by the reference; this pins the format itself."""
    from some_amd.utils import smf
    from some_amd.utils.infer_utils import build_midi_file
    seg = {'note_midi': np.array([60.2, 0.0, 63.7], np.float32), 'note_dur': np.array([0.5, 0.25, 1.0], np.float64), 'note_rest': np.array([False, True, False])}
    want = bytes([
        0x4D, 0x54, 0x68, 0x64, 0, 0, 0, 6, 0, 1, 0, 1, 0x01, 0xE0,        # MThd, length 6, format 1, one track, 480 ticks per quarter
        0x4D, 0x54, 0x72, 0x6B, 0, 0, 0, 30,                              # MTrk, 30 bytes follow
        0x00, 0xFF, 0x51, 0x03, 0x07, 0xA1, 0x20,                         # delta 0, set_tempo 500 000 us per quarter (120 bpm)
        0x00, 0x90, 0x3C, 0x40,                                           # delta 0, note_on ch 0, note 60 (round(60.2)), velocity 64
        0x83, 0x60, 0x80, 0x3C, 0x40,                                     # delta 480 (VLQ 83 60), note_off 60
        0x81, 0x70, 0x90, 0x40, 0x40,                                     # delta 240 (the rest), note_on 64 (round(63.7))
        0x87, 0x40, 0x80, 0x40, 0x40,                                     # delta 960 (VLQ 87 40), note_off 64
        0x00, 0xFF, 0x2F, 0x00])                                          # delta 0, end_of_track
    assert build_midi_file([0.0], [seg], tempo=120).to_bytes() == want
    # running status: consecutive channel messages with the same status byte drop it (what mido's writer does); a meta event resets it
    tr = smf.MidiTrack([smf.Message('note_on', note=60, time=0), smf.Message('note_on', note=64, time=0), smf.MetaMessage('set_tempo', tempo=500000, time=0),
                        smf.Message('note_on', note=67, time=1), smf.Message('note_off', note=60, time=0x4000)])
    mf = smf.MidiFile()
    mf.tracks.append(tr)
    body = mf.to_bytes()[22:]
    assert body == bytes([0x00, 0x90, 0x3C, 0x40, 0x00, 0x40, 0x40, 0x00, 0xFF, 0x51, 0x03, 0x07, 0xA1, 0x20, 0x01, 0x90, 0x43, 0x40,
                          0x81, 0x80, 0x00, 0x80, 0x3C, 0x40, 0x00, 0xFF, 0x2F, 0x00])


def test_smf_variable_length_quantity():
    from some_amd.utils.smf import _vlq
    assert _vlq(0) == [0] and _vlq(0x7F) == [0x7F] and _vlq(0x80) == [0x81, 0x00]
    assert _vlq(0x3FFF) == [0xFF, 0x7F] and _vlq(0x4000) == [0x81, 0x80, 0x00] and _vlq(480 * 8) == [0x9E, 0x00]


def test_wav_roundtrip_and_rate_check(tmp_path):
    from some_amd.utils.audio import load_wav, save_wav
    y = synth.synth_clip(1, 0.2)
    save_wav(tmp_path / 'a.wav', y, 44100)
    z, sr = load_wav(tmp_path / 'a.wav', 44100)
    assert sr == 44100 and z.dtype == np.float32 and np.abs(z - y).max() <= 1 / 32768 + 1e-7


@pytest.mark.parametrize('file_sr', [48000, 22050, 32000])
def test_wav_loader_resamples_like_librosa_kaiser_best(tmp_path, file_sr):
    """librosa.load(path, sr=44100) accepts any source rate (infer.py:34, batch_infer.py:51).  The restated resampy
    'kaiser_best' interpolation is checked against the analytic answer (a sum of sinusoids is its own band-limited
    interpolant) and against scipy's polyphase resampler; length follows librosa's fix_length(ceil(n * ratio))."""
    from scipy.signal import resample_poly
    from some_amd.utils.audio import load_pcm, load_wav, resample, save_wav
    sr, secs = 44100, 0.5
    freqs, amps = (220.0, 1234.5, 5000.0), (0.3, 0.2, 0.1)

    def tone(rate):
        t = np.arange(int(secs * rate)) / rate
        return sum(a * np.sin(2 * np.pi * f * t) for f, a in zip(freqs, amps)).astype(np.float32)

    y = resample(tone(file_sr), file_sr, sr)
    assert y.dtype == np.float32 and len(y) == int(np.ceil(int(secs * file_sr) * sr / file_sr))
    want = tone(sr)
    n = min(len(y), len(want))
    edge = 2000                                           # the 64-zero-crossing filter is truncated at the ends
    assert np.abs(y[edge:n - edge] - want[edge:n - edge]).max() < 2e-4
    import math
    g = math.gcd(sr, file_sr)
    poly = resample_poly(tone(file_sr).astype(np.float64), sr // g, file_sr // g)
    assert np.abs(y[edge:n - edge] - poly[edge:n - edge]).max() < 2e-3
    # through the file loaders (int16 PCM on disk): both entry points resample, stereo is averaged first
    save_wav(tmp_path / 'a.wav', tone(file_sr), file_sr)
    z, got_sr = load_wav(tmp_path / 'a.wav', sr)
    assert got_sr == sr and z.dtype == np.float32 and np.abs(z[edge:n - edge] - want[edge:n - edge]).max() < 5e-4
    z2, _ = load_pcm(tmp_path / 'a.wav', sr)
    np.testing.assert_array_equal(z, z2)
    assert resample(want, sr, sr) is not None and np.array_equal(resample(want, sr, sr), want)


def test_load_pcm_fast_path_equals_scipy_and_falls_back(tmp_path):
    """load_pcm's two-syscall reader (mono int16 PCM at the target rate: what DiffSinger datasets hold) returns scipy's samples bit for
    bit; every other layout - stereo, float, another rate, an extensible header, extra chunks in front of the data, a truncated file -
    takes the scipy / load_wav path and gives load_wav's result."""
    import struct
    from scipy.io import wavfile
    from some_amd.utils import audio
    rng = np.random.default_rng(3)
    sr = 44100
    pcm = rng.integers(-32768, 32767, size=12345, dtype=np.int16)
    wavfile.write(str(tmp_path / 'mono.wav'), sr, pcm)
    fast = audio._read_pcm16_mono(tmp_path / 'mono.wav', sr)
    assert fast is not None and fast.dtype == np.int16 and np.array_equal(fast, pcm)
    got, got_sr = audio.load_pcm(tmp_path / 'mono.wav', sr)
    assert got_sr == sr and got.dtype == np.int16 and np.array_equal(got, pcm)
    # a LIST chunk between fmt and data (what many editors write) is skipped, odd chunk sizes are word aligned
    raw = (tmp_path / 'mono.wav').read_bytes()
    extra = b'LIST' + struct.pack('<I', 5) + b'abcde' + b'\0'
    body = raw[12:36] + extra + raw[36:]
    (tmp_path / 'list.wav').write_bytes(b'RIFF' + struct.pack('<I', 4 + len(body)) + b'WAVE' + body)
    assert np.array_equal(audio._read_pcm16_mono(tmp_path / 'list.wav', sr), pcm)
    assert np.array_equal(audio.load_pcm(tmp_path / 'list.wav', sr)[0], pcm)
    # fallbacks
    wavfile.write(str(tmp_path / 'stereo.wav'), sr, np.stack([pcm, pcm[::-1]], axis=1))
    wavfile.write(str(tmp_path / 'float.wav'), sr, (pcm / 32768.0).astype(np.float32))
    wavfile.write(str(tmp_path / 'rate.wav'), 22050, pcm[:2000])
    for name in ('stereo.wav', 'float.wav', 'rate.wav'):
        assert audio._read_pcm16_mono(tmp_path / name, sr) is None
        want, _ = audio.load_wav(tmp_path / name, sr, mono=True)
        got, _ = audio.load_pcm(tmp_path / name, sr)
        assert got.dtype == np.float32 and np.array_equal(got, want)
    # pooled sample buffers: same samples, the buffer is reused after give(), a truncated file hands its buffer back
    pool = audio.PcmPool(max_bytes=8 << 20)
    a, _ = audio.load_pcm(tmp_path / 'mono.wav', sr, pool)
    assert np.array_equal(a, pcm) and a.base is not None and a.base.shape[0] == audio.PcmPool.STEP
    first = a.base
    pool.give(a)
    b, _ = audio.load_pcm(tmp_path / 'list.wav', sr, pool)
    assert b.base is first and np.array_equal(b, pcm)
    c, _ = audio.load_pcm(tmp_path / 'mono.wav', sr, pool)                 # pool exhausted? no: a second block fits in 8 MiB
    d, _ = audio.load_pcm(tmp_path / 'mono.wav', sr, pool)
    e, _ = audio.load_pcm(tmp_path / 'mono.wav', sr, pool)                 # 4 blocks x 2 MiB = the cap; the 5th is a plain array
    f, _ = audio.load_pcm(tmp_path / 'mono.wav', sr, pool)
    assert f.base is None and np.array_equal(f, pcm) and pool.bytes == 8 << 20
    pool.give(f)                                                           # a foreign array is simply dropped
    for x in (b, c, d, e):
        pool.give(x)
    assert sum(len(v) for v in pool._free.values()) == 4 and not pool._owned
    (tmp_path / 'short.wav').write_bytes(raw[:-100])                      # data chunk longer than the file
    assert audio._read_pcm16_mono(tmp_path / 'short.wav', sr) is None
    (tmp_path / 'junk.wav').write_bytes(b'not a wave file at all, just bytes' * 4)
    assert audio._read_pcm16_mono(tmp_path / 'junk.wav', sr) is None


def test_config_inheritance(tmp_path, monkeypatch):
    from some_amd.utils.config_utils import read_full_config
    monkeypatch.chdir(tmp_path)
    (tmp_path / 'configs').mkdir()
    (tmp_path / 'configs/base.yaml').write_text('a: 1\nargs: {x: 1, y: 2}\n')
    (tmp_path / 'configs/mid.yaml').write_text('base_config: configs/base.yaml\nb: 2\nargs: {y: 3}\n')
    (tmp_path / 'configs/top.yaml').write_text('base_config:\n  - configs/mid.yaml\nc: 3\n')
    cfg = read_full_config(tmp_path / 'configs/top.yaml')
    assert cfg == {'a': 1, 'b': 2, 'c': 3, 'args': {'x': 1, 'y': 3}}


def test_train_config_chains_resolve_like_the_reference(tmp_path, monkeypatch):
    """train.py --config FILE: recursive base_config chains through read_full_config (utils/config_utils.py:19-41, train.py:33-35),
    the built-in configs addressable as configs/<name>.yaml, unknown keys kept, and NO fallback to a default model."""
    import sys
    import click
    from some_amd.configs import get_config
    from some_amd.utils import config_utils
    root = __import__('pathlib').Path(__file__).resolve().parents[1]
    sys.path.insert(0, str(root))
    import train
    monkeypatch.chdir(tmp_path)
    config_utils._loaded.clear()
    assert train._load_config('two_head_model') == get_config('two_head_model')
    assert train._load_config('configs/quant_two_head_model.yaml') == get_config('quant_two_head_model')
    (tmp_path / 'exp').mkdir()
    (tmp_path / 'exp/mid.yaml').write_text('base_config: configs/two_head_model.yaml\nmax_batch_frames: 1234\nmy_own_key: {a: 1}\n'
                                          'midi_extractor_args: {lay: 5}\n')
    (tmp_path / 'exp/top.yaml').write_text('base_config:\n  - mid.yaml\noptimizer_args: {lr: 0.5}\n')       # relative to the including file
    cfg = train._load_config(str(tmp_path / 'exp/top.yaml'))
    want = get_config('two_head_model', lay=5)
    want.update(max_batch_frames=1234, my_own_key={'a': 1})
    want['optimizer_args']['lr'] = 0.5
    assert cfg == want
    # a model the built-ins do not know (the reference's continuous.yaml / discrete.yaml, a typo): an error, not two_head_model
    (tmp_path / 'exp/other.yaml').write_text('base_config: configs/continuous.yaml\n')
    with pytest.raises(FileNotFoundError, match='continuous.yaml'):
        train._load_config(str(tmp_path / 'exp/other.yaml'))
    (tmp_path / 'exp/partial.yaml').write_text('base_config: configs/base.yaml\nmax_batch_size: 2\n')
    with pytest.raises(click.UsageError, match='task_cls'):
        train._load_config(str(tmp_path / 'exp/partial.yaml'))
    # a real configs/ tree on disk wins over the built-ins (the reference's own files, a user's edits)
    (tmp_path / 'configs').mkdir()
    (tmp_path / 'configs/two_head_model.yaml').write_text('base_config: configs/base.yaml\nfrom_disk: true\n')
    (tmp_path / 'configs/base.yaml').write_text('hop_size: 256\n')
    config_utils._loaded.clear()
    assert config_utils.read_full_config('configs/two_head_model.yaml') == {'hop_size': 256, 'from_disk': True}


def test_cpu_device_is_refused(tmp_path):
    import inference
    from some_amd.configs import get_config
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        inference.MIDIExtractionInference(config=get_config('midi_conformer', lay=1), model_path=tmp_path / 'x.ckpt', device='cpu')


def test_extraction_service_dispatch_logic(tmp_path):
    """The service's queue / coalescing / error-string logic with the model stubbed (no GPU): every request gets its
    own result, requests that queued up together share one ``infer_files`` call, handler errors are strings."""
    import threading
    import dataset_util
    from some_amd.configs import get_config
    from some_amd.serving import ExtractionService
    from some_amd.utils.audio import save_wav

    calls = []

    class Stub(dataset_util.FakeIngestInference):
        def infer_files(self, files, slicer):
            calls.append(len(files))
            gate.wait(timeout=30)
            return super().infer_files(files, slicer)

    gate = threading.Event()
    files = [np.clip(np.round(synth.synth_clip(500 + i, 1.0 + i).astype(np.float64) * 32768), -32768, 32767).astype(np.int16)
             for i in range(6)]
    (tmp_path / 'm.ckpt').write_bytes(b'')
    with ExtractionService(work_dir=tmp_path) as svc:
        svc._instances[str(svc.resolve_model('m.ckpt'))] = (Stub(), get_config('midi_conformer'))
        futs = [svc.submit('m.ckpt', files[0])]           # occupies the dispatcher (blocked on the gate) ...
        while not calls:
            pass
        futs += [svc.submit('m.ckpt', f) for f in files[1:]]      # ... while these queue up
        gate.set()
        res = [f.result(timeout=60) for f in futs]
        assert calls == [1, 5] and svc.batches_run == 2 and svc.requests_served == 6
        ref = dataset_util.FakeIngestInference()
        from some_amd.utils.slicer2 import Slicer
        for f, r in zip(files, res):
            want = ref.infer_files([f], Slicer(sr=44100, max_sil_kept=1000))[0]
            assert len(r) == len(want)
            for (o1, a), (o2, b) in zip(r, want):
                assert o1 == o2 and np.array_equal(a['note_dur'], b['note_dur'])
        wav = tmp_path / 'a.wav'
        save_wav(wav, synth.synth_clip(1, 2.0), 44100)
        mid, msg = svc.extract_midi('m.ckpt', wav, 120, output_midi_path=tmp_path / 'o.mid')
        assert mid.read_bytes()[:4] == b'MThd' and msg.startswith('Cost ')
        assert svc.extract_midi(None, wav, 120)[1] == 'Error: required inputs not specified.'
        (tmp_path / 'bad.wav').write_bytes(b'xx')
        assert svc.extract_midi('m.ckpt', tmp_path / 'bad.wav', 120)[0] is None
        with pytest.raises(ValueError):
            svc.submit('m.ckpt', np.zeros((2, 10), dtype=np.float32))
    with pytest.raises(RuntimeError):
        svc.submit('m.ckpt', files[0])


def test_align_pool_matches_inline(tmp_path):
    """The alignment worker subprocesses return exactly what ``align_job`` returns in-process."""
    import dataset_util
    from some_amd import batch_logic
    from some_amd.align_worker import AlignPool
    fake = dataset_util.FakeInference()
    jobs = {}
    for i in range(12):
        segs = fake.infer([np.zeros((300 + 40 * i) * 512, np.float32), np.zeros(200 * 512, np.float32)])
        jobs[i] = ([0.0, 4.0 + i], segs, ' '.join(['0.400000'] * 20), ' '.join(['2'] * 10), bool(i % 2))
    pool = AlignPool(3)
    for k, a in jobs.items():
        pool.submit(k, a)
    got = pool.close()
    assert got == {k: batch_logic.align_job(*a) for k, a in jobs.items()}
    bad = AlignPool(1)
    bad.submit('x', ([0.0], [{'note_midi': np.zeros(2, np.float32), 'note_dur': np.zeros(1), 'note_rest': np.zeros(2, bool)}], '0.5', '1', False))
    with pytest.raises(AssertionError):
        bad.close()


def test_arena_cache_file_validation(tmp_path, monkeypatch):
    """The cached flat weight file is used only while checkpoint size / mtime, arena size, precision and the model shape
    keys all match; anything else (or a truncated file) reads as 'no cache'."""
    from some_amd import arena_cache
    from some_amd.configs import get_config
    cfg = get_config('midi_conformer', lay=1)
    ckpt = tmp_path / 'm.ckpt'
    ckpt.write_bytes(b'x' * 100)
    arena = np.arange(1000, dtype=np.float32)
    assert arena_cache.load(ckpt, 1000, 1, cfg) is None
    assert arena_cache.store(ckpt, arena, 1, cfg)
    np.testing.assert_array_equal(arena_cache.load(ckpt, 1000, 1, cfg), arena)
    assert arena_cache.load(ckpt, 1000, 0, cfg) is None                       # other precision: other file
    assert arena_cache.load(ckpt, 999, 1, cfg) is None                        # arena size changed (library layout)
    assert arena_cache.load(ckpt, 1000, 1, get_config('midi_conformer', lay=2)) is None
    path = arena_cache.cache_path(ckpt, 1)
    path.write_bytes(path.read_bytes()[:-8])                                  # truncated
    assert arena_cache.load(ckpt, 1000, 1, cfg) is None
    assert arena_cache.store(ckpt, arena, 1, cfg)
    ckpt.write_bytes(b'y' * 101)                                              # checkpoint replaced
    assert arena_cache.load(ckpt, 1000, 1, cfg) is None
    ro = tmp_path / 'missing_dir' / 'm.ckpt'
    assert arena_cache.store(ro, arena, 1, cfg) is False                      # unwritable: best effort
    # a rebuilt libsome_amd.so (possibly another arena layout / folding) must not be handed an old cache file
    assert arena_cache.store(ckpt, arena, 1, cfg) and arena_cache.load(ckpt, 1000, 1, cfg) is not None
    monkeypatch.setattr(arena_cache, '_library_crc', lambda: 0x12345678)
    assert arena_cache.load(ckpt, 1000, 1, cfg) is None


def test_http_front_of_the_service(tmp_path):
    """/models lists checkpoints, /infer turns WAV bytes into MIDI bytes through ExtractionService (model stubbed)."""
    import warnings
    import dataset_util
    from some_amd.configs import get_config
    from some_amd.serve import build_app
    from some_amd.serving import ExtractionService
    from some_amd.utils.audio import save_wav
    warnings.simplefilter('ignore')
    from starlette.testclient import TestClient
    (tmp_path / 'exp').mkdir()
    (tmp_path / 'exp' / 'm.ckpt').write_bytes(b'')
    wav = tmp_path / 'a.wav'
    save_wav(wav, synth.synth_clip(2, 2.0), 44100)
    with ExtractionService(work_dir=tmp_path) as svc:
        svc._instances[str(svc.resolve_model('exp/m.ckpt'))] = (dataset_util.FakeIngestInference(), get_config('midi_conformer'))
        client = TestClient(build_app(svc, tmp_path))
        assert client.get('/models').json() == {'models': ['exp/m.ckpt']}
        # only checkpoints inside the work directory are served (webui.py:82-88 offers a closed list): no absolute paths,
        # no '..', no other suffixes - and different spellings of one file share one cached instance
        outside = tmp_path.parent / 'outside.ckpt'
        outside.write_bytes(b'')
        for bad in (str(outside), '../outside.ckpt', 'exp/../../outside.ckpt', 'exp/config.yaml', 'exp/nope.ckpt', '/etc/passwd'):
            r = client.post('/infer', params={'model': bad}, content=wav.read_bytes())
            assert r.status_code == 404 and 'unknown model' in r.json()['error'], bad
            assert svc.extract_midi(bad, wav, 120) == (None, f'Error: unknown model: {bad}')
        r = client.post('/infer', params={'model': './exp/../exp/m.ckpt', 'tempo': 100}, content=wav.read_bytes())
        assert r.status_code == 200 and len(svc._instances) == 1
        r = client.post('/infer', params={'model': 'exp/m.ckpt', 'tempo': 100}, content=wav.read_bytes())
        assert r.status_code == 200 and r.content[:4] == b'MThd' and r.headers['x-some-stats'].startswith('Cost ')
        r = client.post('/infer', params={'model': 'exp/m.ckpt'}, content=b'junk')
        assert r.status_code == 400 and 'unsupported or corrupt' in r.json()['error']
        r = client.post('/infer', params={'model': 'exp/m.ckpt'}, content=b'')
        assert r.status_code == 400
