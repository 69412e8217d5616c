#!/bin/bash
# round 5 (last GPU seconds): hardware queues per process for the four-stream training step
O=gpurun_out; mkdir -p $O
for spec in "queues=4(default)|X=0" "queues=8|GPU_MAX_HW_QUEUES=8"; do
  name=${spec%%|*}; envs=${spec#*|}
  echo -n "$name frames=520: "
  env $envs timeout 40 python tools/train_bench.py --mixed --frames 520 --steps 60 --warmup 10 2>&1 | tail -1 | sed 's/two_head_model lay 3 (mixed bf16): //'
done | tee $O/r05ao_train_hwq.txt
