#!/bin/bash
# small-batch A/B: tools/exp_batch.sh <tag> "name|ENV=.." ...   (ms per step at 1 / 2 / 4 / 8 clips of 30 s, dual-stream forward)
TAG=$1; shift
O=gpurun_out; mkdir -p $O
for B in 1 2 4 8; do
  for spec in "$@"; do
    name=${spec%%|*}; envs=${spec#*|}
    env $envs timeout 300 python bench.py --batch $B --steps 40 --warmup 10 --no-cpu-baseline --no-latency --no-f32-leg --no-secondary --no-kernel-profile > $O/${TAG}_${name}_b$B.json 2> $O/${TAG}_${name}_b$B.err
    python -c "
import json
d=json.loads(open('$O/${TAG}_${name}_b$B.json').read().strip().splitlines()[-1]); print('B=$B', '$name', d['ms_per_step'], d.get('notes_decoded_last_step'))" 2>&1 | tail -1
  done
done
