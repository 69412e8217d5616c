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
