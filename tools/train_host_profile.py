"""Where the HOST time of a training step goes (cProfile over the trainer's enqueue path; kernels run asynchronously).

    python tools/train_host_profile.py [--frames 520] [--steps 30]"""
import argparse
import cProfile
import pathlib
import pstats
import sys

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from some_amd.configs import get_config  # noqa: E402
from some_amd.training.task import MIDIExtractionTrainer  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--frames', type=int, default=520)
ap.add_argument('--batch', type=int, default=8)
ap.add_argument('--steps', type=int, default=30)
ap.add_argument('--top', type=int, default=45)
a = ap.parse_args()
tr = MIDIExtractionTrainer(get_config('two_head_model', pl_trainer_precision='bf16'), device='cuda:0', seed=1)
B, T = a.batch, a.frames
rng = np.random.default_rng(0)
u2n = np.repeat(np.arange(1, T // 40 + 2), 40)[:T][None].repeat(B, 0)
sample = {'units': torch.from_numpy((rng.standard_normal((B, T, 80)) - 4).astype(np.float32)).cuda(), 'unit2note': torch.from_numpy(u2n).cuda(),
          'probs': torch.rand(B, T, 128, device='cuda') * 0.1, 'bounds': (torch.from_numpy(np.diff(u2n, axis=1, prepend=0)) > 0).float().cuda()}
for _ in range(5):
    tr.training_step(sample)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(a.steps):
    tr.training_step(sample)
pr.disable()
st = pstats.Stats(pr)
st.sort_stats('tottime').print_stats(a.top)
