"""Persistent service (webui.py:21-66 handler semantics without Gradio): concurrent callers are coalesced into packed
device batches; results equal direct ``infer_files`` calls.  Needs a real MI355X."""
import threading

import numpy as np
import pytest

from some_amd import synth
from some_amd.configs import get_config

pytestmark = pytest.mark.gpu


def _pcm(w):
    return np.clip(np.round(w.astype(np.float64) * 32768.0), -32768, 32767).astype(np.int16)


def test_concurrent_requests_are_batched_and_match_direct_calls(tmp_path):
    from some_amd.serving import ExtractionService
    from some_amd.utils.audio import save_wav
    from some_amd.utils.slicer2 import Slicer
    cfg = get_config('midi_conformer', lay=1)
    synth.save_checkpoint(cfg, tmp_path / 'exp' / 'model.ckpt', seed=9)
    files = [_pcm(synth.synth_clip(300 + i, 2.0 + 1.7 * i, silence_every=3.0 if i % 2 else 0.0)) for i in range(10)]
    with ExtractionService(work_dir=tmp_path) as svc:
        first = svc.submit('exp/model.ckpt', files[0]).result(timeout=600)        # loads the model
        results = [None] * len(files)

        def worker(i):
            results[i] = svc.submit('exp/model.ckpt', files[i]).result(timeout=600)

        threads = [threading.Thread(target=worker, args=(i,)) for i in range(len(files))]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        assert svc.requests_served == len(files) + 1
        assert svc.batches_run < svc.requests_served                      # some requests shared a device batch
        ins, _ = svc._instances[str(svc.resolve_model('exp/model.ckpt'))]
        assert len(svc._instances) == 1                                   # './exp/../exp/model.ckpt' style spellings share it
        slicer = Slicer(sr=44100, max_sil_kept=1000)
        for i, f in enumerate(files):
            direct = ins.infer_files([f], slicer)[0]
            assert [off for off, _ in results[i]] == [off for off, _ in direct]
            for (_, a), (_, b) in zip(results[i], direct):
                # packing-dependent rounding only (key tiles are aligned in packed-batch coordinates): same notes
                np.testing.assert_array_equal(a['note_dur'], b['note_dur'])
                np.testing.assert_array_equal(a['note_rest'], b['note_rest'])
                np.testing.assert_allclose(a['note_midi'], b['note_midi'], rtol=0, atol=1e-3)
        for (_, a), (_, b) in zip(first, results[0]):
            np.testing.assert_array_equal(a['note_dur'], b['note_dur'])

        # the web handler: file in, MIDI file + statistics string out; errors are strings, not exceptions
        wav = tmp_path / 'song.wav'
        save_wav(wav, synth.synth_clip(400, 7.0, silence_every=3.0), 44100)
        mid, msg = svc.extract_midi('exp/model.ckpt', wav, 100)
        assert mid == wav.with_suffix('.mid') and mid.read_bytes()[:4] == b'MThd' and msg.startswith('Cost ') and 'RTF' in msg
        assert svc.extract_midi('', wav, 100) == (None, 'Error: required inputs not specified.')
        bad = tmp_path / 'bad.wav'
        bad.write_bytes(b'not a wav file')
        assert svc.extract_midi('exp/model.ckpt', bad, 100) == (None, 'Error: unsupported or corrupt file format: bad.wav')
    with pytest.raises(RuntimeError):
        svc.submit('exp/model.ckpt', files[0])
