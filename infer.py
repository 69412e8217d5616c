#!/usr/bin/env python
"""``python infer.py --model CKPT --wav WAV [--midi OUT.mid] [--tempo 120]`` - the reference's command line
(infer.py:14-19): WAV -> silence slicer -> packed GPU batch -> notes -> Standard MIDI File."""
import pathlib

import click

from some_amd.inference.loader import load_inference   # noqa: F401  (re-exported: batch_infer.py and tools import it from here)
from some_amd.utils.audio import load_pcm
from utils.infer_utils import build_midi_file
from utils.slicer2 import Slicer


def extract_to_midi(infer_ins, config: dict, wav_path: pathlib.Path, tempo: float):
    """infer.py:34-38 of the reference (librosa.load -> Slicer.slice -> infer -> build_midi_file); the file is uploaded as
    stored and sliced from the RMS curve computed on the device (same chunk boundaries)."""
    rate = config['audio_sample_rate']
    samples, _ = load_pcm(wav_path, sr=rate)
    segments = infer_ins.infer_files([samples], Slicer(sr=rate, max_sil_kept=1000))[0]
    return build_midi_file([offset for offset, _ in segments], [notes for _, notes in segments], tempo=tempo)


@click.command(help='Run inference with a trained model')
@click.option('--model', required=True, metavar='CKPT_PATH', help='Path to the model checkpoint (*.ckpt)')
@click.option('--wav', required=True, metavar='WAV_PATH', help='Path to the input wav file (*.wav)')
@click.option('--midi', required=False, metavar='MIDI_PATH', help='Path to the output MIDI file (*.mid)')
@click.option('--tempo', required=False, type=float, default=120, metavar='TEMPO', help='Specify tempo in the output MIDI')
def infer(model, wav, midi, tempo):
    infer_ins, config = load_inference(pathlib.Path(model))
    source = pathlib.Path(wav)
    target = pathlib.Path(midi) if midi is not None else source.with_suffix('.mid')
    extract_to_midi(infer_ins, config, source, tempo).save(target)
    print(f'MIDI file saved at: \'{target}\'')


if __name__ == '__main__':
    infer()
