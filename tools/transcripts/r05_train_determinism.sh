#!/bin/bash
# round 5: run-to-run determinism probe of the training step at the reference's batch shape (separate processes, same seeds)
O=gpurun_out; mkdir -p $O
for spec in "lanes=1|SOME_AMD_TRAIN_LANES=1" "lanes=2|SOME_AMD_TRAIN_LANES=2" "lanes=2+wgrad-lanes|SOME_AMD_TRAIN_WGRAD_LANES=1" "lanes=1,calls|SOME_AMD_TRAIN_LANES=1 SOME_AMD_TRAIN_BLOCK_CALLS=0 SOME_AMD_TRAIN_DEVICE_PREP=0"; do
  name=${spec%%|*}; envs=${spec#*|}
  for rep in 1 2 3; do
    echo -n "$name: "
    env $envs timeout 200 python tools/train_bench.py --mixed --frames 520 --steps 6 --warmup 0 --digest 2>&1 | grep digest
  done
done | tee $O/r05aj_train_determinism.txt
