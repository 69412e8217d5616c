#!/bin/bash
# round 5: training-step A/B at the reference's batch shape (8 x 520 frames) and at 8 x 2584; switches by environment
O=gpurun_out; mkdir -p $O; TAG=$1; shift
for rep in 1 2; do
for spec in "$@"; do
  name=${spec%%|*}; envs=${spec#*|}
  for fr in 520 2584; do
    echo -n "$name frames=$fr: "
    env $envs timeout 300 python tools/train_bench.py --mixed --frames $fr --steps 20 --warmup 5 2>&1 | tail -1
  done
done; done | tee $O/${TAG}_train_ab.txt
