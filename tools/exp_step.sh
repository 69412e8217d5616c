#!/bin/bash
# step-time-only A/B legs (dual-stream forward, no per-kernel leg): tools/exp_step.sh <tag> <repeats> "name|ENV=.." ...
TAG=$1; REP=$2; shift 2
O=gpurun_out; mkdir -p $O
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-latency --no-f32-leg --no-secondary --no-kernel-profile --no-e2e --no-train"
for r in $(seq $REP); do
  for spec in "$@"; do
    name=${spec%%|*}; envs=${spec#*|}
    env $envs timeout 300 $B > $O/${TAG}_${name}_$r.json 2> $O/${TAG}_${name}_$r.err
    python -c "
import json,sys
d=json.loads(open('$O/${TAG}_${name}_$r.json').read().strip().splitlines()[-1]); print('$name', $r, d['ms_per_step'], d.get('notes_decoded_last_step'))" 2>&1 | tail -1
  done
done
