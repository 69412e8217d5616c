#!/usr/bin/env python
"""``python infer.py --model CKPT --wav WAV [--midi OUT.mid] [--tempo 120]`` - same command line as the
reference's infer.py:14-19; WAV -> chunks (Slicer) -> packed GPU batch -> notes -> Standard MIDI File."""
import importlib
import pathlib

import click
import yaml

import inference
from some_amd.utils.audio import load_pcm
from utils.config_utils import print_config
from utils.infer_utils import build_midi_file
from utils.slicer2 import Slicer


def load_inference(model_path: pathlib.Path, device=None):
    """infer.py:20-31 / batch_infer.py:21-34: config.yaml beside the checkpoint -> task -> inference class."""
    with open(model_path.with_name('config.yaml'), 'r', encoding='utf8') as f:
        config = yaml.safe_load(f)
    print_config(config)
    infer_cls = inference.task_inference_mapping[config['task_cls']]
    pkg, cls_name = infer_cls.rsplit('.', 1)
    infer_cls = getattr(importlib.import_module(pkg), cls_name)
    assert issubclass(infer_cls, inference.BaseInference), \
        f'Inference class {infer_cls} is not a subclass of {inference.BaseInference}.'
    return infer_cls(config=config, model_path=model_path, device=device), config


@click.command(help='Run inference with a trained model')
@click.option('--model', required=True, metavar='CKPT_PATH', help='Path to the model checkpoint (*.ckpt)')
@click.option('--wav', required=True, metavar='WAV_PATH', help='Path to the input wav file (*.wav)')
@click.option('--midi', required=False, metavar='MIDI_PATH', help='Path to the output MIDI file (*.mid)')
@click.option('--tempo', required=False, type=float, default=120, metavar='TEMPO', help='Specify tempo in the output MIDI')
def infer(model, wav, midi, tempo):
    model_path = pathlib.Path(model)
    infer_ins, config = load_inference(model_path)
    wav_path = pathlib.Path(wav)
    # infer.py:34-38 of the reference (librosa.load -> Slicer.slice -> infer -> build_midi_file); the file is uploaded
    # as stored and sliced from the RMS curve computed on the device (same chunk boundaries)
    samples, _ = load_pcm(wav_path, sr=config['audio_sample_rate'])
    slicer = Slicer(sr=config['audio_sample_rate'], max_sil_kept=1000)
    segments = infer_ins.infer_files([samples], slicer)[0]
    midi_file = build_midi_file([off for off, _ in segments], [seg for _, seg in segments], tempo=tempo)
    midi_path = pathlib.Path(midi) if midi is not None else wav_path.with_suffix('.mid')
    midi_file.save(midi_path)
    print(f'MIDI file saved at: \'{midi_path}\'')


if __name__ == '__main__':
    infer()
