#!/bin/bash
# round 5: depthwise-convolution parameter gradients into the gradient arrays (and onto the weight-gradient side streams): parity + A/B
O=gpurun_out; mkdir -p $O
( SOME_AMD_TRAIN_DWCONV_SINKS=1 timeout 200 python -m pytest tests/test_gpu_train_step.py -x -q -m gpu -k "depthwise or weight_gradient_lanes or tape_equals or two_lanes_equal or replicas or matches_reference or host_sync" 2>&1 | tail -6 ) > $O/r05am_pytest_dwconv_sinks.txt
tail -2 $O/r05am_pytest_dwconv_sinks.txt
for rep in 1 2; do
for spec in "tape-adds|SOME_AMD_TRAIN_DWCONV_SINKS=0" "sinks|SOME_AMD_TRAIN_DWCONV_SINKS=1"; do
  name=${spec%%|*}; envs=${spec#*|}
  echo -n "$name frames=520: "
  env $envs timeout 100 python tools/train_bench.py --mixed --frames 520 --steps 60 --warmup 10 2>&1 | tail -1 | sed 's/two_head_model lay 3 (mixed bf16): //'
done; done | tee $O/r05am_train_ab.txt
