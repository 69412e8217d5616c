"""Golden-vector generator - TEST INFRASTRUCTURE ONLY.  Runs ONLY in the build container (needs /root/reference).

Executes the REFERENCE'S OWN Python modules (imported from /root/reference, never copied) on seeded inputs
and stores their outputs as small fixtures under tests/golden/.  The GPU box has no /root/reference, so
the fixtures are what travels.  Loader strategy (verified here):

* ``modules.model.Gmidi_conform`` (+ conform / attention / conv) import as-is (torch + einops only).
* ``modules/rmvpe/spec.py``, ``utils/infer_utils.py``, ``utils/slicer2.py``, ``batch_infer.py`` are loaded
  file-level (their packages' ``__init__`` pull in lightning / torchaudio, absent here) with stubs for the
  absent third-party modules: ``librosa.filters.mel`` / ``librosa.midi_to_note`` (-> the restatements in
  oracle/restate.py and some_amd/utils - "parity unpinned" third-party arithmetic), ``mido`` (recording stub).

Usage:  python oracle/make_golden.py            (from the repo root)
"""
import importlib.util
import json
import os
import pathlib
import sys
import types

import numpy as np
import torch

REPO = pathlib.Path(__file__).resolve().parents[1]
REF = pathlib.Path('/root/reference')
OUT = pathlib.Path(os.environ['SOME_GOLDEN_OUT']) if os.environ.get('SOME_GOLDEN_OUT') else REPO / 'tests' / 'golden'   # override: regenerate elsewhere and diff

sys.path.insert(0, str(REF))          # reference packages win name clashes (modules/, utils/, inference/)
sys.path.append(str(REPO))

from oracle import restate  # noqa: E402
from some_amd import synth  # noqa: E402
from some_amd.configs import get_config  # noqa: E402


def _load_file(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


# ---- stubs for absent third-party modules ---------------------------------------------------------
NOTE_NAMES = ['C', 'C#', 'D', 'D#', 'E', 'F', 'F#', 'G', 'G#', 'A', 'A#', 'B']


def _midi_to_note(midi, unicode=False):
    n = int(np.round(midi))
    return f'{NOTE_NAMES[n % 12]}{int(n // 12) - 1}'


librosa = types.ModuleType('librosa')
librosa.filters = types.ModuleType('librosa.filters')
librosa.filters.mel = lambda sr, n_fft, n_mels, fmin, fmax, htk: restate.mel_filterbank(sr, n_fft, n_mels, fmin, fmax)
librosa.midi_to_note = _midi_to_note
librosa.load = None
sys.modules['librosa'] = librosa
sys.modules['librosa.filters'] = librosa.filters


class _Msg:
    def __init__(self, kind, **kw):
        self.kind, self.kw = kind, kw


mido = types.ModuleType('mido')
mido.MidiFile = lambda charset='utf8': types.SimpleNamespace(tracks=[])
mido.MidiTrack = list
mido.MetaMessage = lambda kind, **kw: _Msg(kind, **kw)
mido.Message = lambda kind, **kw: _Msg(kind, **kw)
mido.bpm2tempo = lambda bpm: int(round(60 * 1e6 / bpm))
sys.modules['mido'] = mido

ref_spec = _load_file('ref_spec', REF / 'modules/rmvpe/spec.py')
ref_infer_utils = _load_file('ref_infer_utils', REF / 'utils/infer_utils.py')
ref_slicer2 = _load_file('ref_slicer2', REF / 'utils/slicer2.py')
from modules.model.Gmidi_conform import midi_conforms as RefModel  # noqa: E402


def ref_model(config, seed):
    import copy
    model = RefModel(copy.deepcopy(config)).eval()
    sd = synth.synth_state_dict(config, seed)
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    return model


def ref_mel(config):
    return ref_spec.MelSpectrogram(
        n_mel_channels=config['units_dim'], sampling_rate=config['audio_sample_rate'],
        win_length=config['win_size'], hop_length=config['hop_size'],
        mel_fmin=config['fmin'], mel_fmax=config['fmax'])


def gen_mel():
    cfg = get_config('midi_conformer')
    mel = ref_mel(cfg)
    out = {}
    rng = np.random.default_rng(7)
    cases = {
        'clip0_1s': synth.synth_clip(0, 1.0),
        'clip1_odd': synth.synth_clip(1, 0.37)[:16001],
        'tiny100': (rng.standard_normal(100) * 0.1).astype(np.float32),
        'zeros3000': np.zeros(3000, dtype=np.float32),
        'noise_hop': (rng.standard_normal(512 * 9) * 0.05).astype(np.float32),
    }
    for k, w in cases.items():
        with torch.no_grad():
            u = mel(torch.from_numpy(w)[None]).transpose(1, 2)[0].contiguous().numpy()
        out[k + '.audio'] = w
        out[k + '.units'] = u
    out['mel_basis'] = mel.mel_basis.numpy()
    np.savez_compressed(OUT / 'mel.npz', **out)
    print('mel.npz', {k: v.shape for k, v in out.items()})


SHIFT_CASES = [
    # name, keyshift, speed, center  (me_binarizer.py:237-239 draws float or rounded key shifts in [-12, 12])
    ('up12', 12, 1, True),
    ('down12', -12, 1, True),
    ('up5', 5, 1, True),
    ('down3p7', -3.7, 1, True),
    ('up0p31', 0.31, 1, True),
    ('speed1p3', 0, 1.3, True),
    ('shift_speed', 7, 0.8, True),
    ('nocenter', 0, 1, False),
    ('nocenter_down2', -2, 1, False),
]


def gen_mel_shift():
    cfg = get_config('midi_conformer')
    mel = ref_mel(cfg)
    wav = synth.synth_clip(3, 1.2)
    out = {'audio': wav}
    for name, ks, sp, ce in SHIFT_CASES:
        with torch.no_grad():
            out[name] = mel(torch.from_numpy(wav)[None], keyshift=ks, speed=sp, center=ce).transpose(1, 2)[0].contiguous().numpy()
    np.savez_compressed(OUT / 'mel_shift.npz', **out)
    print('mel_shift.npz', {k: v.shape for k, v in out.items()})


MODEL_CASES = [
    # name, config, lay, seed, B, T, masked
    ('conf_lay8', 'midi_conformer', 8, 11, 1, 173, False),
    ('conf_lay2_b2', 'midi_conformer', 2, 12, 2, 96, False),
    ('quant_lay3', 'quant_two_head_model', 3, 13, 1, 131, False),
    ('two_head_lay1_mask', 'two_head_model', 1, 14, 1, 64, True),
    ('conf_lay1_t1', 'midi_conformer', 1, 15, 1, 1, False),
    ('conf_lay1_t33', 'midi_conformer', 1, 16, 1, 33, False),
]


def gen_model():
    out = {}
    meta = {}
    for name, cname, lay, seed, b, t, masked in MODEL_CASES:
        cfg = get_config(cname, lay=lay)
        model = ref_model(cfg, seed)
        rng = np.random.default_rng(seed + 100)
        units = (rng.standard_normal((b, t, cfg['units_dim'])) * 2.0 - 4.0).astype(np.float32)
        mask = np.ones((b, t), dtype=bool)
        if masked:
            mask[:, t - 9:] = False
            mask[:, 5] = False
        quant = cname.startswith('quant')
        with torch.no_grad():
            x, m = torch.from_numpy(units), torch.from_numpy(mask)
            logits, bounds = model(x, None, mask=m)
            probs, bounds2 = model(x, None, mask=m, softmax=quant, sig=not quant)
        assert torch.equal(bounds, bounds2)
        out[name + '.units'] = units
        out[name + '.mask'] = mask
        out[name + '.logits'] = logits.numpy()
        out[name + '.probs'] = probs.numpy()
        out[name + '.bounds'] = bounds.numpy()
        meta[name] = dict(config=cname, lay=lay, seed=seed, B=b, T=t, quant=quant)
    np.savez_compressed(OUT / 'model.npz', **out)
    (OUT / 'model.json').write_text(json.dumps(meta, indent=1))
    print('model.npz', list(meta))


def gen_decode():
    out = {}
    cfg = get_config('midi_conformer')
    iu = ref_infer_utils
    # (1) the reference's only textual known-answer (utils/infer_utils.py:103-113, commented __main__)
    f2i = torch.LongTensor([[1, 1, 1, 1, 2, 2, 3, 3, 3, 0, 0, 0, 0, 0], [1, 1, 1, 2, 3, 3, 3, 3, 3, 4, 4, 0, 0, 0]])
    vals = torch.FloatTensor([[60, 61, 60.5, 63, 57, 57, 50, 55, 54, 0, 0, 0, 0, 0],
                              [50, 51, 50.5, 53, 47, 47, 40, 45, 44, 38, 38, 0, 0, 0]])
    iv, idur, im = iu.decode_note_sequence(f2i, vals, f2i > 0)
    out['kat.frame2item'], out['kat.values'] = f2i.numpy(), vals.numpy()
    out['kat.item_values'], out['kat.item_dur'], out['kat.item_masks'] = iv.numpy(), idur.numpy(), im.numpy()
    # (2) seeded random continuous / quantised cases, one clip each (B=1 as the reference runs them)
    for ci, (t, nb, seed) in enumerate([(400, 128, 1), (862, 128, 2), (37, 128, 3), (500, 129, 4), (2584, 129, 5), (1, 128, 6), (2584, 128, 7)]):
        rng = np.random.default_rng(seed)
        quant = nb == 129
        # note-like probs: a gaussian bump that moves, with some rests
        centers = np.repeat(rng.uniform(40, 80, t // 20 + 1), 20)[:t] + rng.standard_normal(t) * 0.3
        idx = np.arange(nb)[None, :]
        bump = np.exp(-0.5 * (idx - centers[:, None]) ** 2)
        amp = rng.uniform(0.02, 1.0, (t, 1))
        logits = (np.log(bump * amp + 1e-4) + rng.standard_normal((t, nb)) * 0.2).astype(np.float32)
        if quant:
            probs = torch.softmax(torch.from_numpy(logits), dim=-1)
        else:
            probs = torch.sigmoid(torch.from_numpy(logits))
        bounds = torch.from_numpy((rng.uniform(0, 1, t) ** 6).astype(np.float32))
        masks = torch.ones(1, t, dtype=torch.bool)
        p, b = probs[None].clone(), bounds[None].clone()
        p *= masks[..., None]
        b *= masks
        f2i = iu.decode_bounds_to_alignment(b) * masks
        if quant:
            midi = p.argmax(dim=-1)
            rest = midi == 128
            v = midi.clip(min=0, max=127)
        else:
            v, rest = iu.decode_gaussian_blurred_probs(p, vmin=cfg['midi_min'], vmax=cfg['midi_max'],
                                                       deviation=cfg['midi_prob_deviation'], threshold=cfg['rest_threshold'])
        nm, nd, nmask = iu.decode_note_sequence(f2i, v, ~rest & masks)
        k = f'case{ci}'
        out[k + '.probs'], out[k + '.bounds'] = probs.numpy(), bounds.numpy()
        out[k + '.frame2item'], out[k + '.values'], out[k + '.rest'] = f2i[0].numpy(), v[0].numpy(), rest[0].numpy()
        out[k + '.note_midi'] = nm[0].numpy()
        out[k + '.note_dur'] = nd[0].numpy() * (cfg['hop_size'] / cfg['audio_sample_rate'])
        out[k + '.note_dur_frames'] = nd[0].numpy()
        out[k + '.note_rest'] = (~nmask)[0].numpy()
        out[k + '.quant'] = np.array(quant)
    np.savez_compressed(OUT / 'decode.npz', **out)
    print('decode.npz', len(out))


SCALED = dict(midi_min=36.0, midi_max=96.5, midi_prob_deviation=0.8, rest_threshold=0.05)   # interval 60.5 / 127


def gen_decode_scaled():
    """A non-default value range: idx * interval + vmin is no longer exact in fp32, so a fused multiply-add in the
    decoder would show (tests/golden/decode_scaled.npz)."""
    iu = ref_infer_utils
    t, nb = 600, 128
    rng = np.random.default_rng(8)
    centers = np.repeat(rng.uniform(10, 110, t // 20 + 1), 20)[:t] + rng.standard_normal(t) * 0.3
    bump = np.exp(-0.5 * ((np.arange(nb)[None, :] - centers[:, None]) / 1.7) ** 2)
    logits = (np.log(bump * rng.uniform(0.02, 1.0, (t, 1)) + 1e-4) + rng.standard_normal((t, nb)) * 0.2).astype(np.float32)
    probs = torch.sigmoid(torch.from_numpy(logits))
    bounds = torch.from_numpy((rng.uniform(0, 1, t) ** 6).astype(np.float32))
    masks = torch.ones(1, t, dtype=torch.bool)
    p, b = probs[None].clone(), bounds[None].clone()
    f2i = iu.decode_bounds_to_alignment(b) * masks
    v, rest = iu.decode_gaussian_blurred_probs(p, vmin=SCALED['midi_min'], vmax=SCALED['midi_max'],
                                               deviation=SCALED['midi_prob_deviation'], threshold=SCALED['rest_threshold'])
    nm, nd, nmask = iu.decode_note_sequence(f2i, v, ~rest & masks)
    out = {'probs': probs.numpy(), 'bounds': bounds.numpy(), 'frame2item': f2i[0].numpy(), 'values': v[0].numpy(),
           'rest': rest[0].numpy(), 'note_midi': nm[0].numpy(), 'note_dur_frames': nd[0].numpy(), 'note_rest': (~nmask)[0].numpy()}
    out.update({k: np.array(x) for k, x in SCALED.items()})
    np.savez_compressed(OUT / 'decode_scaled.npz', **out)
    print('decode_scaled.npz', len(out))


def gen_slicer():
    out = {}
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
    meta = {}
    for k, w in cases.items():
        chunks = ref_slicer2.Slicer(sr=44100, max_sil_kept=1000).slice(w)
        meta[k] = [[float(c['offset']), int(c['waveform'].shape[0])] for c in chunks]
        rms = ref_slicer2.get_rms(y=w, frame_length=3528, hop_length=882).squeeze(0) if len(w) > 882 * 250 else np.zeros(0, np.float32)
        out[k + '.rms'] = rms.astype(np.float32)
    (OUT / 'slicer.json').write_text(json.dumps(meta, indent=1))
    np.savez_compressed(OUT / 'slicer_rms.npz', **out)
    print('slicer.json', {k: len(v) for k, v in meta.items()})


def gen_midi_msgs():
    """build_midi_file (utils/infer_utils.py:79-100) through the recording mido stub."""
    rng = np.random.default_rng(21)
    cases = {}
    for ci in range(3):
        offsets, segs = [], []
        t0 = 0.0
        for s in range(3):
            n = int(rng.integers(3, 9))
            seg = {'note_midi': rng.uniform(45, 80, n).astype(np.float32),
                   'note_dur': rng.integers(5, 90, n).astype(np.int64) * (512 / 44100),
                   'note_rest': rng.uniform(0, 1, n) < 0.25}
            offsets.append(t0)
            # make the 2nd case overrun the next chunk offset to exercise the clamp at :92-93
            t0 += float(seg['note_dur'].sum()) * (0.8 if ci == 1 else 1.2)
            segs.append(seg)
        mf = ref_infer_utils.build_midi_file(offsets, segs, tempo=[120, 97.5, 140][ci])
        msgs = [[m.kind, int(m.kw.get('note', -1)), int(m.kw.get('time', 0)), int(m.kw.get('tempo', -1))] for m in mf.tracks[0]]
        cases[f'case{ci}'] = {
            'tempo': [120, 97.5, 140][ci], 'offsets': offsets,
            'segments': [{k: v.tolist() for k, v in s.items()} for s in segs], 'messages': msgs}
    (OUT / 'midi_msgs.json').write_text(json.dumps(cases))
    print('midi_msgs.json')


def gen_batch_infer_fns():
    """Pure-Python helpers of batch_infer.py (:37-46, :84-134) loaded file-level with stubbed imports."""
    for name in ('inference', 'utils', 'utils.config_utils', 'utils.slicer2'):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules['inference'].BaseInference = object
    sys.modules['utils.config_utils'].print_config = lambda c: None
    sys.modules['utils.slicer2'].Slicer = ref_slicer2.Slicer
    bi = _load_file('ref_batch_infer', REF / 'batch_infer.py')
    rng = np.random.default_rng(33)
    cases = []
    for ci in range(6):
        n_words = int(rng.integers(4, 12))
        ph_num = rng.integers(1, 4, n_words).tolist()
        ph_dur = [round(float(x), 6) for x in rng.uniform(0.05, 0.6, int(sum(ph_num)))]
        words = bi.get_word_durs(ph_dur, ph_num)
        total = words[-1][1]
        notes, t = [], 0.0
        while t < total + 0.3:
            d = round(float(rng.uniform(0.08, 0.9)), 6)
            midi = float(np.float32(rng.uniform(48, 76)))
            rest = bool(rng.uniform() < 0.2)
            st = round(t + (0.03 if rng.uniform() < 0.3 else 0.0), 6)
            notes.append({'start_time': st, 'end_time': round(st + d, 6), 'note_seq': bi.calc_seq(midi, rest),
                          'note_dur': d, '_midi': midi, '_rest': rest})
            t = notes[-1]['end_time']
        import copy
        aligned = bi.midi_align(copy.deepcopy(notes), words)
        per_word = []
        for w in words:
            per_word.append({'max': bi.get_max_overlap_midi(w, aligned),
                             'all': [s['note_seq'] + '@' + repr(s['start_time']) for s in bi.get_all_overlap_midis(w, aligned)]})
        cases.append({'ph_dur': ph_dur, 'ph_num': ph_num, 'words': words, 'notes': notes,
                      'aligned': aligned, 'per_word': per_word})
    seqs = [[float(np.float32(m)), bi.calc_seq(float(np.float32(m)), False)] for m in
            [60.0, 60.25, 59.75, 61.5, 62.5, 69.004, 68.996, 0.4, 127.0, 47.51, 47.49, 71.995]]
    (OUT / 'batch_infer_fns.json').write_text(json.dumps({'cases': cases, 'calc_seq': seqs}))
    print('batch_infer_fns.json')


def gen_batch_csv():
    """The reference's whole batch_infer command (batch_infer.py:149-226) on a miniature dataset, with the
    model replaced by tests/dataset_util.FakeInference and librosa.load by a WAV reader: pins the CSV text."""
    import tempfile
    sys.path.append(str(REPO / 'tests'))
    import dataset_util
    from some_amd.utils.audio import load_wav
    for name in ('inference', 'utils', 'utils.config_utils', 'utils.slicer2'):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules['inference'].BaseInference = object
    sys.modules['utils.config_utils'].print_config = lambda c: None
    sys.modules['utils.slicer2'].Slicer = ref_slicer2.Slicer
    librosa.load = lambda path, sr, mono: load_wav(path, sr=sr, mono=mono)
    bi = _load_file('ref_batch_infer2', REF / 'batch_infer.py')
    cfg = get_config('midi_conformer')
    bi.model_init = lambda p: (dataset_util.FakeInference(), cfg)
    with tempfile.TemporaryDirectory() as d:
        d = pathlib.Path(d)
        dataset_util.build_dataset(d)
        for tag, rm in (('round', True), ('full', False)):
            out = d / f'out_{tag}.csv'
            bi.batch_infer.callback(dataset=str(d), model=str(d / 'm.ckpt'), round_midi=rm, csv=str(out), overwrite=True)
            (OUT / f'batch_csv_{tag}.csv').write_bytes(out.read_bytes())
    print('batch_csv_*.csv')


def gen_deploy():
    """Deployment twin (deployment/base_onnx_module.py:38-79, me_onnx_module.py:23-39): reflect-padded STFT front
    end and the waveform -> notes forward, executed with the reference's own classes (B = 2)."""
    sys.modules.setdefault('utils', types.ModuleType('utils'))
    sys.modules['utils'].build_object_from_class_name = lambda *a, **k: None
    sys.modules.setdefault('utils.infer_utils', ref_infer_utils)
    base = _load_file('ref_base_onnx', REF / 'deployment/base_onnx_module.py')
    cfg = get_config('midi_conformer', lay=2)
    mel = base.MelSpectrogram_ONNX(n_mel_channels=80, sampling_rate=44100, win_length=2048, hop_length=512, mel_fmin=40, mel_fmax=8000)
    model = ref_model(cfg, 41)
    iu = ref_infer_utils
    w = np.stack([synth.synth_clip(70, 2.0), synth.synth_clip(71, 2.0)])
    out = {}
    with torch.no_grad():
        units = mel(torch.from_numpy(w)).transpose(1, 2)
        out['units'] = units.numpy().copy()
        masks = torch.ones(units.shape[:2], dtype=torch.bool)
        probs, bounds = model(x=units, f0=None, mask=masks, sig=True)
        out['probs'], out['bounds'] = probs.numpy().copy(), bounds.numpy().copy()
        f2i = iu.decode_bounds_to_alignment(bounds, use_diff=False) * masks
        v, rest = iu.decode_gaussian_blurred_probs(probs, vmin=0, vmax=127, deviation=1.0, threshold=0.1)
        nm, nd, nmask = iu.decode_note_sequence(f2i, v, ~rest & masks)
    # me_onnx_module.py:39: `note_dur_pred * self.timestep` is a TORCH op: int64 tensor * python float -> float32
    out['note_midi'], out['note_rest'], out['note_dur'] = nm.numpy(), (~nmask).numpy(), (nd * (512 / 44100)).numpy()
    np.savez_compressed(OUT / 'deploy.npz', **out)
    print('deploy.npz', {k: v.shape for k, v in out.items()})


def gen_train():
    """One training step of the reference model + losses (training/me_task.py:96-109) with torch autograd and
    torch.optim.AdamW: losses, a digest of every parameter gradient, BatchNorm running stats, parameters after the step.
    Dropout probabilities are 0 (torch's RNG stream is not reproducible elsewhere); BatchNorm runs in train mode."""
    import copy
    import zlib
    cfg = get_config('two_head_model', lay=1)
    for k in ('conv_drop', 'ffn_latent_drop', 'ffn_out_drop', 'attention_drop'):
        cfg['midi_extractor_args'][k] = 0.0
    model = RefModel(copy.deepcopy(cfg)).train()
    sd = synth.synth_state_dict(cfg, 31)
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    ref_losses = _load_file('ref_bound_loss', REF / 'modules/losses/bound_loss.py')
    batch = synth.synth_train_batch()
    t = {k: torch.from_numpy(v) for k, v in batch.items()}
    opt = torch.optim.AdamW(model.parameters(), lr=1e-4 * (1 / 5000), betas=(0.9, 0.98), weight_decay=0)   # WarmupLR at step 1
    mask = t['unit2note'] > 0
    probs, bounds = model(x=t['units'], f0=None, mask=mask, sig=False)
    bound_loss = ref_losses.BinaryEMDLoss()(bounds, t['bounds'])
    midi_loss = torch.nn.BCEWithLogitsLoss()(probs, t['probs'])
    (bound_loss + midi_loss).backward()
    out = {'bound_loss': np.array(bound_loss.item()), 'midi_loss': np.array(midi_loss.item()),
           'probs_head': probs.detach().numpy()[:, :4, :8].copy(), 'bounds_out': bounds.detach().numpy().copy()}
    names = []
    for name, p in model.named_parameters():
        g = p.grad.detach().numpy().astype(np.float64).reshape(-1)
        proj = np.random.default_rng(zlib.crc32(name.encode())).standard_normal(g.size)
        out['grad.' + name] = np.array(list(g[:8]) + [0.0] * max(0, 8 - g.size) + [g.sum(), np.abs(g).sum(), np.sqrt((g * g).sum()), (g * proj).sum()])
        out['sk.' + name] = grad_sketch(name, g)
        names.append(name)
    # Lightning's gradient_clip_val = clip_grad_norm (configs/base.yaml:49, train.py:88), algorithm 'norm'
    out['grad_norm'] = np.array(float(torch.nn.utils.clip_grad_norm_(model.parameters(), cfg.get('clip_grad_norm', 1.0))))
    opt.step()
    for name, p in model.named_parameters():
        v = p.detach().numpy().astype(np.float64).reshape(-1)
        out['after.' + name] = np.array(list(v[:8]) + [0.0] * max(0, 8 - v.size) + [v.sum()])
    for name, b in model.named_buffers():
        if name.endswith('running_mean') or name.endswith('running_var'):
            out['buf.' + name] = b.detach().numpy().copy()
    out['names'] = np.array(names)
    np.savez_compressed(OUT / 'train_step.npz', **out)
    print('train_step.npz', len(out), 'bound_loss', bound_loss.item(), 'midi_loss', midi_loss.item())


def _grad_digest(name, p):
    import zlib
    g = p.grad.detach().float().numpy().astype(np.float64).reshape(-1)
    proj = np.random.default_rng(zlib.crc32(name.encode())).standard_normal(g.size)
    return np.array(list(g[:8]) + [0.0] * max(0, 8 - g.size) + [g.sum(), np.abs(g).sum(), np.sqrt((g * g).sum()), (g * proj).sum()])


def grad_sketch(name, g, buckets=64):
    """64-bucket count sketch of a gradient tensor (seeded signs, element i -> bucket i % 64): ||sketch(a) - sketch(b)|| estimates
    ||a - b|| to about 10 % - a whole-tensor error measure that fits in a fixture (tests/test_gpu_train_step.py recomputes it)."""
    import zlib
    g = np.asarray(g, dtype=np.float64).reshape(-1)
    signs = np.random.default_rng(zlib.crc32(name.encode()) + 1).integers(0, 2, g.size) * 2.0 - 1.0
    return np.bincount(np.arange(g.size) % buckets, weights=g * signs, minlength=buckets)


def gen_train_bf16():
    """The SAME step as gen_train (same weights, batch, losses) run the way Lightning's ``precision='bf16'`` runs it
    (train.py:65, configs/midi_conformer.yaml:35): forward and losses under ``torch.autocast(dtype=torch.bfloat16)``, fp32 master
    weights, backward through the recorded dtypes.  Stored: the losses and the per-parameter gradient digests of gen_train - the
    yardstick for the HIP bf16 path: how far the REFERENCE'S OWN bf16 arithmetic sits from its fp32 step, tensor by tensor."""
    import copy
    cfg = get_config('two_head_model', lay=1)
    for k in ('conv_drop', 'ffn_latent_drop', 'ffn_out_drop', 'attention_drop'):
        cfg['midi_extractor_args'][k] = 0.0
    model = RefModel(copy.deepcopy(cfg)).train()
    sd = synth.synth_state_dict(cfg, 31)
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    ref_losses = _load_file('ref_bound_loss', REF / 'modules/losses/bound_loss.py')
    t = {k: torch.from_numpy(v) for k, v in synth.synth_train_batch().items()}
    mask = t['unit2note'] > 0
    with torch.autocast('cpu', dtype=torch.bfloat16):
        probs, bounds = model(x=t['units'], f0=None, mask=mask, sig=False)
        bound_loss = ref_losses.BinaryEMDLoss()(bounds, t['bounds'])
        midi_loss = torch.nn.BCEWithLogitsLoss()(probs, t['probs'])
    (bound_loss + midi_loss).backward()
    out = {'bound_loss': np.array(float(bound_loss)), 'midi_loss': np.array(float(midi_loss)),
           'probs_dtype': np.array(str(probs.dtype)), 'bounds_dtype': np.array(str(bounds.dtype))}
    names = []
    for name, p in model.named_parameters():
        out['grad.' + name] = _grad_digest(name, p)
        out['sk.' + name] = grad_sketch(name, p.grad.detach().float().numpy())
        names.append(name)
    out['grad_norm'] = np.array(float(torch.nn.utils.clip_grad_norm_(model.parameters(), cfg.get('clip_grad_norm', 1.0))))
    out['names'] = np.array(names)
    np.savez_compressed(OUT / 'train_step_bf16.npz', **out)
    print('train_step_bf16.npz', len(out), 'bound_loss', float(bound_loss), 'midi_loss', float(midi_loss), probs.dtype, bounds.dtype)


TRAJ_STEPS = 8
TRAJ_PARAMS = ('model.inln.weight', 'model.cf_lay.1.att2.ffn2.ln1.weight', 'model.cf_lay.2.att1.att.to_out.0.weight')


def gen_train_trajectory():
    """TRAJ_STEPS consecutive fp32 updates of the configs[4] model (two_head_model: lay 3) through the reference's own pieces:
    ``midi_conforms`` + BinaryEMDLoss + BCEWithLogitsLoss (training/me_task.py:79-111), ``torch.optim.AdamW`` built as
    training/base_task.py:331-344 builds it, the reference's ``WarmupLR`` stepped once per update (base_task.py:346-358:
    interval 'step'), ``clip_grad_norm_`` (train.py:88).  A DIFFERENT batch every step; warm-up shortened to 4 updates so the
    window covers the linear ramp, the peak and the 1/sqrt decay and the weights really move (lr up to 1e-4).  Dropout is 0 (torch's
    mask stream cannot be reproduced elsewhere; the HIP dropout is gated statistically in tests/test_gpu_train_ops.py).
    Stored per step: both losses, the pre-clip gradient norm, the rate used, digests of three parameters after the update, and the
    BatchNorm running statistics + num_batches_tracked of one conv module; at the end a digest of every parameter."""
    import copy
    cfg = get_config('two_head_model')
    assert cfg['midi_extractor_args']['lay'] == 3
    for k in ('conv_drop', 'ffn_latent_drop', 'ffn_out_drop', 'attention_drop'):
        cfg['midi_extractor_args'][k] = 0.0
    model = RefModel(copy.deepcopy(cfg)).train()
    sd = synth.synth_state_dict(cfg, 47)
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    ref_losses = _load_file('ref_bound_loss', REF / 'modules/losses/bound_loss.py')
    sched_mod = _load_file('ref_scheduler', REF / 'lr_scheduler/scheduler.py')
    oa = cfg['optimizer_args']
    opt = torch.optim.AdamW(model.parameters(), lr=oa['lr'], betas=(oa['beta1'], oa['beta2']), weight_decay=oa['weight_decay'])
    sched = sched_mod.WarmupLR(opt, warmup_steps=4, min_lr=cfg['lr_scheduler_args']['min_lr'])
    params = dict(model.named_parameters())
    bufs = dict(model.named_buffers())
    bn = [k for k in bufs if k.endswith('running_mean')][2]                     # one conv module's BatchNorm
    bn_var, bn_cnt = bn.replace('running_mean', 'running_var'), bn.replace('running_mean', 'num_batches_tracked')
    for k in TRAJ_PARAMS:
        assert k in params, (k, list(params)[:40])
    out = {'steps': np.array(TRAJ_STEPS), 'warmup_steps': np.array(4), 'weights_seed': np.array(47), 'bn_name': np.array(bn),
           'traj_params': np.array(TRAJ_PARAMS)}
    rows = {k: [] for k in ('bound_loss', 'midi_loss', 'grad_norm', 'lr', 'bn_cnt')}
    for step in range(TRAJ_STEPS):
        t = {k: torch.from_numpy(v) for k, v in synth.synth_train_batch(B=2 + step % 2, T=80 + 16 * (step % 3), seed=100 + step).items()}
        mask = t['unit2note'] > 0
        opt.zero_grad(set_to_none=True)
        probs, bounds = model(x=t['units'], f0=None, mask=mask, sig=False)
        bound_loss = ref_losses.BinaryEMDLoss()(bounds, t['bounds'])
        midi_loss = torch.nn.BCEWithLogitsLoss()(probs, t['probs'])
        (bound_loss + midi_loss).backward()
        rows['bound_loss'].append(bound_loss.item())
        rows['midi_loss'].append(midi_loss.item())
        rows['grad_norm'].append(float(torch.nn.utils.clip_grad_norm_(model.parameters(), cfg.get('clip_grad_norm', 1.0))))
        rows['lr'].append(opt.param_groups[0]['lr'])
        opt.step()
        sched.step()
        rows['bn_cnt'].append(int(bufs[bn_cnt]))
        for k in TRAJ_PARAMS:
            v = params[k].detach().numpy().astype(np.float64).reshape(-1)
            out[f'p{step}.{k}'] = np.array(list(v[:8]) + [v.sum(), np.sqrt((v * v).sum())])
        out[f'bn{step}.mean'] = bufs[bn].detach().numpy().copy()
        out[f'bn{step}.var'] = bufs[bn_var].detach().numpy().copy()
    # the same 8 updates the way Lightning's precision='bf16' runs them (train.py:65, configs/midi_conformer.yaml:35): forward and losses under
    # torch.autocast(bfloat16), fp32 master weights and optimiser - the yardstick of the HIP bf16 trajectory (how far the REFERENCE's own bf16
    # arithmetic drifts from its fp32 run over 8 updates)
    model16 = RefModel(copy.deepcopy(cfg)).train()
    model16.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    opt16 = torch.optim.AdamW(model16.parameters(), lr=oa['lr'], betas=(oa['beta1'], oa['beta2']), weight_decay=oa['weight_decay'])
    sched16 = sched_mod.WarmupLR(opt16, warmup_steps=4, min_lr=cfg['lr_scheduler_args']['min_lr'])
    rows16 = {k: [] for k in ('bound_loss', 'midi_loss', 'grad_norm')}
    for step in range(TRAJ_STEPS):
        t = {k: torch.from_numpy(v) for k, v in synth.synth_train_batch(B=2 + step % 2, T=80 + 16 * (step % 3), seed=100 + step).items()}
        opt16.zero_grad(set_to_none=True)
        with torch.autocast('cpu', dtype=torch.bfloat16):
            probs, bounds = model16(x=t['units'], f0=None, mask=t['unit2note'] > 0, sig=False)
            bound_loss = ref_losses.BinaryEMDLoss()(bounds, t['bounds'])
            midi_loss = torch.nn.BCEWithLogitsLoss()(probs, t['probs'])
        (bound_loss + midi_loss).backward()
        rows16['bound_loss'].append(float(bound_loss))
        rows16['midi_loss'].append(float(midi_loss))
        rows16['grad_norm'].append(float(torch.nn.utils.clip_grad_norm_(model16.parameters(), cfg.get('clip_grad_norm', 1.0))))
        opt16.step()
        sched16.step()
    for k, v in rows16.items():
        out['bf16.' + k] = np.array(v)
    for k, v in rows.items():
        out[k] = np.array(v)
    names = []
    for name, p in params.items():
        v = p.detach().numpy().astype(np.float64).reshape(-1)
        out['final.' + name] = np.array(list(v[:8]) + [0.0] * max(0, 8 - v.size) + [v.sum(), np.sqrt((v * v).sum())])
        out['init.' + name] = np.array([np.sqrt((np.asarray(sd[name], dtype=np.float64) ** 2).sum())])
        names.append(name)
    out['names'] = np.array(names)
    np.savez_compressed(OUT / 'train_trajectory.npz', **out)
    print('train_trajectory.npz', len(out), 'losses', rows['bound_loss'], rows['midi_loss'], 'lr', rows['lr'], 'norm', rows['grad_norm'])
    print('   autocast bf16 arm:', rows16['bound_loss'], rows16['midi_loss'])


def synth_quant_batch(B=2, T=96, seed=33):
    """Items shaped like the quantised binarizer's (preprocessing/me_quant_binarizer.py:12-32): units, pitch, integer note classes
    (128 = rest), note durations in frames, the 1-based frame -> note map; the second item is shorter (padding in the batch)."""
    rng = np.random.default_rng(seed)
    items = []
    for b in range(B):
        valid = T if b == 0 else T - 26
        edges = np.sort(rng.choice(np.arange(1, valid), size=6, replace=False))
        unit2note = 1 + np.searchsorted(edges, np.arange(valid), side='right')
        note_midi = rng.integers(40, 80, size=7)
        note_midi[rng.integers(0, 7)] = 128
        items.append({'units': (rng.standard_normal((valid, 80)) * 1.5 - 4.0).astype(np.float32), 'pitch': np.zeros(valid, np.float32),
                      'note_midi': note_midi.astype(np.int64), 'note_dur': np.bincount(unit2note, minlength=8)[1:].astype(np.int64),
                      'unit2note': unit2note.astype(np.int64)})
    return items


def _ref_quant_collater():
    """QuantizedMIDIExtractionDataset.collater (training/me_quant_task.py:14-27) executed from the reference's own source text: the module
    itself cannot be imported here (lightning, torchmetrics), so the method's lines are compiled alone with what they name - torch, F,
    the reference's ``collate_nd`` (utils/__init__.py:25-34, compiled the same way) - and ``super().collater`` (training/base_task.py:73-76:
    ``{'size': len(samples)}``) written out."""
    import ast
    import textwrap
    ns = {'torch': torch, 'F': torch.nn.functional}
    tree = ast.parse((REF / 'utils/__init__.py').read_text())
    fn = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == 'collate_nd')
    exec(compile(ast.Module([fn], []), 'ref_collate_nd', 'exec'), ns)
    src = (REF / 'training/me_quant_task.py').read_text()
    tree = ast.parse(src)
    cls = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == 'QuantizedMIDIExtractionDataset')
    meth = next(n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name == 'collater')
    text = textwrap.dedent(ast.get_source_segment(src, meth)).replace('super().collater(samples)', "{'size': len(samples)}")
    assert "{'size': len(samples)}" in text
    exec(compile(text, 'ref_quant_collater', 'exec'), ns)
    return lambda samples: ns['collater'](None, samples)


def gen_train_quant():
    """One training step of ``QuantizedMIDIExtractionTask`` (training/me_quant_task.py:30-78): the reference model with 129 classes, raw
    logits (softmax=False), nn.CrossEntropyLoss(ignore_index=-1) + BinaryEMDLoss, torch autograd, AdamW - digests as gen_train.  The batch
    comes out of the reference's own collater text (_ref_quant_collater) and is stored, so the GPU test can also pin
    some_amd.training.data.quant_collater against it."""
    import copy
    cfg = get_config('quant_two_head_model', lay=1)
    assert cfg['midi_num_bins'] == 129
    for k in ('conv_drop', 'ffn_latent_drop', 'ffn_out_drop', 'attention_drop'):
        cfg['midi_extractor_args'][k] = 0.0
    model = RefModel(copy.deepcopy(cfg)).train()
    sd = synth.synth_state_dict(cfg, 37)
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    ref_losses = _load_file('ref_bound_loss', REF / 'modules/losses/bound_loss.py')
    items = synth_quant_batch()
    batch = _ref_quant_collater()([{k: torch.from_numpy(v) for k, v in it.items()} for it in items])
    opt = torch.optim.AdamW(model.parameters(), lr=1e-4 * (1 / 10000), betas=(0.9, 0.98), weight_decay=0)   # WarmupLR(10000) at step 1
    mask = batch['unit2note'] > 0
    probs, bounds = model(x=batch['units'], f0=batch['pitch'], mask=mask, softmax=False)
    bound_loss = ref_losses.BinaryEMDLoss(bidirectional=False)(bounds, batch['bounds'])
    midi_loss = torch.nn.CrossEntropyLoss(ignore_index=-1)(probs.transpose(1, 2), batch['midi_idx'])
    (bound_loss + midi_loss).backward()
    out = {'bound_loss': np.array(bound_loss.item()), 'midi_loss': np.array(midi_loss.item()), 'weights_seed': np.array(37)}
    for i, it in enumerate(items):
        for k, v in it.items():
            out[f'item{i}.{k}'] = v
    for k in ('units', 'note_midi', 'note_dur', 'unit2note', 'midi_idx', 'bounds'):
        out['batch.' + k] = batch[k].numpy()
    names = []
    for name, p in model.named_parameters():
        out['grad.' + name] = _grad_digest(name, p)
        out['sk.' + name] = grad_sketch(name, p.grad.detach().numpy())
        names.append(name)
    out['grad_norm'] = np.array(float(torch.nn.utils.clip_grad_norm_(model.parameters(), cfg.get('clip_grad_norm', 1.0))))
    opt.step()
    for name, p in model.named_parameters():
        v = p.detach().numpy().astype(np.float64).reshape(-1)
        out['after.' + name] = np.array(list(v[:8]) + [0.0] * max(0, 8 - v.size) + [v.sum()])
    out['names'] = np.array(names)
    np.savez_compressed(OUT / 'train_step_quant.npz', **out)
    print('train_step_quant.npz', len(out), 'bound_loss', bound_loss.item(), 'midi_loss', midi_loss.item(), 'valid frames', int((batch['midi_idx'] >= 0).sum()))


def gen_lr_schedule():
    """lr_scheduler.scheduler.WarmupLR (the reference's class, on a dummy optimiser) at a few update counts."""
    sched = _load_file('ref_scheduler', REF / 'lr_scheduler/scheduler.py')
    out = {}
    for warmup, min_lr in ((5000, 1e-5), (10, 2e-5), (0, 1e-5)):
        opt = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=1e-4)
        sc = sched.WarmupLR(opt, warmup_steps=warmup, min_lr=min_lr)
        lrs = {}
        for step in range(1, 30001):
            if step in (1, 2, 9, 10, 11, 100, 4999, 5000, 5001, 20000, 30000):
                lrs[str(step)] = opt.param_groups[0]['lr']          # the rate update number `step` is taken with
            opt.step()
            sc.step()
        out[f'{warmup},{min_lr}'] = lrs
    (OUT / 'lr_schedule.json').write_text(json.dumps(out, indent=1))
    print('lr_schedule.json', len(out))


def gen_e2e():
    """waveform -> notes through the reference's own front end, model and decoder (B=1, CPU)."""
    out = {}
    meta = {}
    for name, cname, lay, seed, secs in [('e2e_conf', 'midi_conformer', 2, 31, 4.0), ('e2e_quant', 'quant_two_head_model', 1, 32, 2.5)]:
        cfg = get_config(cname, lay=lay)
        quant = cname.startswith('quant')
        model, mel = ref_model(cfg, seed), ref_mel(cfg)
        w = synth.synth_clip(40 + seed, secs)
        iu = ref_infer_utils
        with torch.no_grad():
            units = mel(torch.from_numpy(w)[None]).transpose(1, 2)
            masks = torch.ones(units.shape[:2], dtype=torch.bool)
            probs, bounds = model(x=units, f0=None, mask=masks, softmax=quant, sig=not quant)
            out[name + '.probs'], out[name + '.bounds'] = probs[0].numpy().copy(), bounds[0].numpy().copy()
            probs *= masks[..., None]
            bounds *= masks
            f2i = iu.decode_bounds_to_alignment(bounds) * masks
            if quant:
                midi = probs.argmax(dim=-1)
                rest = midi == 128
                v = midi.clip(min=0, max=127)
            else:
                v, rest = iu.decode_gaussian_blurred_probs(probs, vmin=0, vmax=127, deviation=1.0, threshold=0.1)
            nm, nd, nmask = iu.decode_note_sequence(f2i, v, ~rest & masks)
        out[name + '.note_midi'] = nm[0].numpy()
        out[name + '.note_dur'] = nd[0].numpy() * (512 / 44100)
        out[name + '.note_rest'] = (~nmask)[0].numpy()
        meta[name] = dict(config=cname, lay=lay, seed=seed, clip=40 + seed, seconds=secs, quant=quant)
    np.savez_compressed(OUT / 'e2e.npz', **out)
    (OUT / 'e2e.json').write_text(json.dumps(meta, indent=1))
    print('e2e.npz')


FULL_CASES = [
    # name, config (BASELINE.json configs[1] / configs[2]), checkpoint seed, first clip seed
    ('full_conf', 'midi_conformer', 1, 0),
    ('full_quant', 'quant_two_head_model', 2, 0),
]
FULL_CLIPS = 8          # 8 x 30 s clips per config (T = 2584 each); probs are stored for clip 0 only (1.3 MB each)


def _ref_training_utils():
    """utils/training_utils.py needs lightning (absent): stub the three names it imports, then import the reference's
    ``utils`` package as it is."""
    if 'lightning' not in sys.modules:
        pl = types.ModuleType('lightning.pytorch')
        pl.LightningModule = object
        cb = types.ModuleType('lightning.pytorch.callbacks')
        cb.ModelCheckpoint = cb.TQDMProgressBar = object
        rz = types.ModuleType('lightning.pytorch.utilities.rank_zero')
        rz.rank_zero_info = print
        ut = types.ModuleType('lightning.pytorch.utilities')
        top = types.ModuleType('lightning')
        top.pytorch, pl.callbacks, pl.utilities, ut.rank_zero = pl, cb, ut, rz
        sys.modules.update({'lightning': top, 'lightning.pytorch': pl, 'lightning.pytorch.callbacks': cb,
                            'lightning.pytorch.utilities': ut, 'lightning.pytorch.utilities.rank_zero': rz})
    # earlier generators (gen_batch_infer_fns / gen_batch_csv / gen_deploy) leave bare stub modules named `utils`, `utils.*` and
    # `inference` in sys.modules ("'utils' is not a package" when the whole file runs in one process): drop every such entry that
    # is not the reference's own file before importing the real package
    for name in [n for n in sys.modules if n == 'utils' or n.startswith('utils.') or n == 'inference']:
        if str(REF) not in str(getattr(sys.modules[name], '__file__', None) or ''):
            del sys.modules[name]
    import utils.training_utils as tu          # the reference's (REF is first on sys.path)
    assert str(REF) in tu.__file__
    return tu


class _Lengths:
    """The three members of BaseDataset the samplers touch (training/base_task.py:55-71)."""

    def __init__(self, sizes):
        self._sizes = np.asarray(sizes)

    def __len__(self):
        return len(self._sizes)

    def num_frames(self, i):
        return self._sizes[i]


def sampler_lengths(n, seed):
    """Frame counts shaped like a sliced singing corpus: log-normal phrase lengths around 6 s, clipped to 1 - 20 s, at
    the 86.13 frames/s of hop 512 @ 44.1 kHz."""
    rng = np.random.default_rng(seed)
    sec = np.clip(rng.lognormal(mean=np.log(6.0), sigma=0.55, size=n), 1.0, 20.0)
    return np.round(sec * 44100 / 512).astype(np.int64)


SAMPLER_CASES = [
    # name, n items, seed, max_batch_frames, max_batch_size, num_replicas, multiple, sort, shuffle_batch, drop_last, grid, epochs
    ('single', 257, 0, 80000, 8, 1, 1, True, False, False, 6, (0, 1)),
    ('ddp8', 1000, 114514, 80000, 8, 8, 1, True, False, False, 6, (0, 3)),
    ('ddp2_accum4', 333, 7, 20000, 48, 2, 4, True, False, False, 200, (0, 2)),
    ('ddp3_unsorted_drop', 100, 1, 9000, 5, 3, 1, False, True, True, 200, (0, 1)),
    ('ddp4_few', 37, 2, 30000, 8, 4, 2, True, True, False, 50, (0, 5)),
]


def gen_samplers():
    tu = _ref_training_utils()
    import utils as ref_utils
    out = {}
    for name, n, seed, mbf, mbs, world, mult, sort, shuf_b, drop, grid, epochs in SAMPLER_CASES:
        ds = _Lengths(sampler_lengths(n, seed + 1))
        plans = {}
        for rank in range(world):
            sm = tu.DsBatchSampler(ds, mbf, mbs, num_replicas=world, rank=rank, frame_count_grid=grid,
                                   required_batch_count_multiple=mult, sort_by_similar_size=sort, shuffle_sample=True,
                                   shuffle_batch=shuf_b, seed=seed, drop_last=drop)
            for ep in epochs:
                sm.set_epoch(ep)
                plans[f'rank{rank}.epoch{ep}'] = [[int(i) for i in b] for b in sm]
        out[name] = {'lengths': ds._sizes.tolist(), 'seed': seed, 'max_batch_frames': mbf, 'max_batch_size': mbs,
                     'num_replicas': world, 'multiple': mult, 'sort': sort, 'shuffle_batch': shuf_b, 'drop_last': drop,
                     'grid': grid, 'plans': plans}
    ds = _Lengths(sampler_lengths(50, 9))
    out['eval'] = {
        'lengths': ds._sizes.tolist(),
        'rank0_fixed': [[int(i) for i in b] for b in tu.DsEvalBatchSampler(ds, 20000, 4, rank=0, batch_by_size=False)],
        'rank0_by_size': [[int(i) for i in b] for b in tu.DsEvalBatchSampler(ds, 3000, 4, rank=0, batch_by_size=True)],
        'rank1': [[int(i) for i in b] for b in tu.DsEvalBatchSampler(ds, 20000, 4, rank=1)],
    }
    out['batch_by_size_multiple'] = [[int(i) for i in b] for b in ref_utils.batch_by_size(
        list(range(50)), ds.num_frames, max_batch_frames=4000, max_batch_size=7, required_batch_size_multiple=2)]
    (OUT / 'samplers.json').write_text(json.dumps(out))
    print('samplers.json', {k: (len(v['plans']) if 'plans' in v else len(v)) for k, v in out.items()})


def gen_fullsize():
    """The BASELINE configs at their OWN size (30 s clips, T = 2584, lay 8 / lay 3): waveform ->
    reference MelSpectrogram -> reference midi_conforms -> reference decode, B = 1 per clip as BaseInference.infer runs
    them (inference/base_infer.py:46-53).  Stored per clip: bounds + decoded notes; probs for clip 0 (and a strided
    sample of every clip's probs) - this is what the HIP path is gated against at the size the metric is quoted on."""
    iu = ref_infer_utils
    out, meta = {}, {}
    for name, cname, seed, clip0 in FULL_CASES:
        cfg = get_config(cname)
        quant = cname.startswith('quant')
        model, mel = ref_model(cfg, seed), ref_mel(cfg)
        for ci in range(FULL_CLIPS):
            w = synth.synth_clip(clip0 + ci, 30.0)
            with torch.no_grad():
                units = mel(torch.from_numpy(w)[None]).transpose(1, 2)
                masks = torch.ones(units.shape[:2], dtype=torch.bool)
                probs, bounds = model(x=units, f0=None, mask=masks, softmax=quant, sig=not quant)
                k = f'{name}.clip{ci}'
                if ci == 0:
                    out[k + '.probs'] = probs[0].numpy().copy()
                    out[k + '.units'] = units[0].numpy().copy()
                out[k + '.probs_s'] = probs[0, 5::37].numpy().copy()          # every 37th frame of every clip
                out[k + '.bounds'] = bounds[0].numpy().copy()
                probs *= masks[..., None]
                bounds *= masks
                f2i = iu.decode_bounds_to_alignment(bounds) * masks
                if quant:
                    midi = probs.argmax(dim=-1)
                    rest = midi == 128
                    v = midi.clip(min=0, max=127)
                else:
                    v, rest = iu.decode_gaussian_blurred_probs(probs, vmin=cfg['midi_min'], vmax=cfg['midi_max'],
                                                               deviation=cfg['midi_prob_deviation'], threshold=cfg['rest_threshold'])
                nm, nd, nmask = iu.decode_note_sequence(f2i, v, ~rest & masks)
            out[k + '.note_midi'] = nm[0].numpy()
            out[k + '.note_dur_frames'] = nd[0].numpy().astype(np.int32)
            out[k + '.note_rest'] = (~nmask)[0].numpy()
            print(name, ci, 'T', units.shape[1], 'notes', nm.shape[1])
        meta[name] = dict(config=cname, lay=cfg['midi_extractor_args']['lay'], seed=seed, clip0=clip0, clips=FULL_CLIPS,
                          seconds=30.0, quant=quant)
    np.savez_compressed(OUT / 'fullsize.npz', **out)
    (OUT / 'fullsize.json').write_text(json.dumps(meta, indent=1))
    print('fullsize.npz')


if __name__ == '__main__':
    torch.manual_seed(0)
    torch.set_num_threads(8)
    OUT.mkdir(parents=True, exist_ok=True)
    if len(sys.argv) > 1:                 # python oracle/make_golden.py gen_mel_shift ...: only the named generators
        for fn in sys.argv[1:]:
            globals()[fn]()
        sys.exit(0)
    gen_mel()
    gen_mel_shift()
    gen_model()
    gen_decode()
    gen_decode_scaled()
    gen_slicer()
    gen_midi_msgs()
    gen_batch_infer_fns()
    gen_batch_csv()
    gen_deploy()
    gen_train()
    gen_train_bf16()
    gen_train_trajectory()
    gen_train_quant()
    gen_lr_schedule()
    gen_e2e()
    gen_fullsize()
    gen_samplers()
