"""Device-side ingest: the slicer's RMS curve and the chunk cut on the GPU.

The RMS must equal numpy's float32 reduction (utils/slicer2.py:5-38) BIT FOR BIT - silence decisions compare it with a
threshold and take argmin - and the whole-file path must give the same chunks and the same notes as
``Slicer.slice`` + ``infer`` on the host.  Needs a real MI355X."""
import numpy as np
import pytest
import torch

from some_amd import synth
from some_amd.configs import get_config
from some_amd.utils.slicer2 import Slicer, get_rms

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def eng():
    from some_amd.engine import Engine
    return Engine(get_config('midi_conformer', lay=0), device='cuda')


def _clips(lengths, seed):
    rng = np.random.default_rng(seed)
    out = []
    for n in lengths:
        y = (rng.standard_normal(n) * 0.05).astype(np.float32)
        if n > 10:
            y[n // 3: n // 2] *= np.float32(1e-3)           # a quiet stretch: values around the -40 dB threshold
        out.append(y)
    return out


@pytest.mark.parametrize('fmt', ['f32', 'pcm16'])
@pytest.mark.parametrize('frame_length,hop,lengths', [
    (3528, 882, [44100 * 3 + 17, 1000, 882, 881, 1, 200000]),
    (2048, 512, [20000, 5, 4096]),
    (100, 30, [1000, 31]),
    (7, 3, [50]),
])
def test_slicer_rms_bitwise_numpy(eng, fmt, frame_length, hop, lengths):
    clips = _clips(lengths, frame_length)
    if fmt == 'pcm16':
        pcm = [np.clip(np.round(c * 32768.0), -32768, 32767).astype(np.int16) for c in clips]
        host = [p.astype(np.float32) / np.float32(32768.0) for p in pcm]       # utils/audio.load_wav
        dev = torch.from_numpy(np.concatenate(pcm)).cuda()
    else:
        host = clips
        dev = torch.from_numpy(np.concatenate(clips)).cuda()
    rms, ro = eng.slicer_rms(dev, [len(c) for c in clips], frame_length, hop)
    got = rms.cpu().numpy()
    for b, y in enumerate(host):
        ref = get_rms(y, frame_length=frame_length, hop_length=hop)[0]
        g = got[ro[b]:ro[b + 1]]
        assert g.shape == ref.shape
        assert np.array_equal(g.view(np.uint32), ref.view(np.uint32)), (b, len(y), np.abs(g - ref).max())


def test_slicer_rms_empty_batch(eng):
    rms, ro = eng.slicer_rms(torch.empty(0, dtype=torch.float32, device='cuda'), [], 3528, 882)
    assert rms.numel() == 0 and ro.tolist() == [0]


@pytest.mark.parametrize('fmt', ['f32', 'pcm16'])
def test_pcm_gather(eng, fmt):
    rng = np.random.default_rng(3)
    n = 300000
    if fmt == 'pcm16':
        src = rng.integers(-32768, 32768, n).astype(np.int16)
        ref_all = src.astype(np.float32) / np.float32(32768.0)
    else:
        src = rng.standard_normal(n).astype(np.float32)
        ref_all = src
    starts = [0, 17, 99999, 250001, 5, n - 1, 1234]
    lens = [1000, 1, 100001, 49999, 0, 1, 4097]
    out, batch = eng.pcm_gather(torch.from_numpy(src).cuda(), starts, lens)
    got = out.cpu().numpy()
    assert batch.B == len(lens) and batch.sample_offsets[-1] == sum(lens)
    assert batch.frame_counts.tolist() == [1 + m // 512 for m in lens]
    for b, (s, m) in enumerate(zip(starts, lens)):
        seg = got[batch.sample_offsets[b]:batch.sample_offsets[b + 1]]
        assert np.array_equal(seg.view(np.uint32), ref_all[s:s + m].view(np.uint32))
    with pytest.raises(ValueError):
        eng.pcm_gather(torch.from_numpy(src).cuda(), [n - 10], [11])


@pytest.mark.parametrize('fmt', ['pcm16', 'f32'])
def test_infer_files_equals_host_slicer_path(tmp_path, fmt):
    """Whole files through the device ingest == Slicer.slice on the host + infer(chunks): same chunk offsets, and
    bit-identical notes (the packed batches are the same, so is every kernel input)."""
    import inference
    cfg = get_config('midi_conformer', lay=1)
    ckpt = synth.save_checkpoint(cfg, tmp_path / 'model.ckpt', seed=5)
    ins = inference.MIDIExtractionInference(config=cfg, model_path=ckpt)
    slicer = Slicer(sr=44100, max_sil_kept=1000)
    waves = [synth.synth_clip(60, 14.0, silence_every=4.0), synth.synth_clip(61, 3.0), synth.synth_clip(62, 9.0, silence_every=3.0),
             synth.synth_clip(63, 6.5)]
    pcm = [np.clip(np.round(w.astype(np.float64) * 32768.0), -32768, 32767).astype(np.int16) for w in waves]
    host = [p.astype(np.float32) / np.float32(32768.0) for p in pcm]
    files = pcm if fmt == 'pcm16' else host
    ins.max_batch_frames = 1500                      # several device batches: exercises the double-buffered staging
    got = ins.infer_files(files, slicer)
    assert len(got) == len(files)
    groups = ins._group_files(files)
    assert len(groups) > 1
    n_chunks = 0
    for idx in groups:                               # same packing on the host path: one device batch per file group
        chunks = [(f, c) for f in idx for c in slicer.slice(host[f])]
        ref = ins.infer_batch([c['waveform'] for _, c in chunks])
        n_chunks += len(chunks)
        pos = 0
        for f in idx:
            mine = [(c, r) for (g, c), r in zip(chunks, ref) if g == f]
            assert len(got[f]) == len(mine)
            for (off, seg), (c, r) in zip(got[f], mine):
                assert off == c['offset']
                for k in ('note_midi', 'note_dur', 'note_rest'):
                    np.testing.assert_array_equal(seg[k], r[k])
            pos += len(mine)
    assert n_chunks > len(files)                     # the slicer did cut something


def test_infer_files_degenerate_lengths(tmp_path):
    """Empty / sub-hop / exactly-one-hop files: one chunk each, T = 1 + L // hop frames, same notes as the host path."""
    import inference
    cfg = get_config('midi_conformer', lay=1)
    ckpt = synth.save_checkpoint(cfg, tmp_path / 'model.ckpt', seed=5)
    ins = inference.MIDIExtractionInference(config=cfg, model_path=ckpt)
    slicer = Slicer(sr=44100, max_sil_kept=1000)
    rng = np.random.default_rng(0)
    files = [np.zeros(0, np.int16), rng.integers(-3000, 3000, 1).astype(np.int16), rng.integers(-3000, 3000, 511).astype(np.int16),
             rng.integers(-3000, 3000, 512).astype(np.int16), rng.integers(-3000, 3000, 5000).astype(np.int16)]
    got = ins.infer_files(files, slicer)
    ref = ins.infer_batch([f.astype(np.float32) / np.float32(32768.0) for f in files])
    for f, g, r in zip(files, got, ref):
        assert len(g) == 1 and g[0][0] == 0
        frames = 1 + len(f) // 512
        assert abs(g[0][1]['note_dur'].sum() - frames * 512 / 44100) < 1e-9
        for k in ('note_midi', 'note_dur', 'note_rest'):
            np.testing.assert_array_equal(g[0][1][k], r[k])
    assert ins.infer_files([], slicer) == []
