#!/bin/bash
# round 5: weight-gradient lanes - parity (the training suite with the lanes forced on) and step-time A/B at the reference's batch shape
O=gpurun_out; mkdir -p $O
export TMPDIR=/tmp
( SOME_AMD_TRAIN_WGRAD_LANES=2 timeout 420 python -m pytest tests/test_gpu_train_step.py tests/test_gpu_train_ffn16.py -x -q -m gpu 2>&1 | tail -15 ) > $O/r05ai_pytest_wgrad_lanes.txt
tail -3 $O/r05ai_pytest_wgrad_lanes.txt
for rep in 1 2; do
for spec in "in-order|SOME_AMD_TRAIN_WGRAD_LANES=0" "lanes|SOME_AMD_TRAIN_WGRAD_LANES=1" "lanes+deferred|SOME_AMD_TRAIN_WGRAD_LANES=2"; do
  name=${spec%%|*}; envs=${spec#*|}
  for fr in 520 2584; do
    echo -n "$name frames=$fr: "
    env $envs timeout 300 python tools/train_bench.py --mixed --frames $fr --steps 30 --warmup 8 2>&1 | tail -1
  done
done; done | tee $O/r05ai_train_ab.txt
SOME_AMD_TRAIN_WGRAD_LANES=2 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r05ai_stats -- python tools/train_bench.py --mixed --frames 520 --steps 25 --warmup 5 > $O/r05ai_stats.log 2>&1
cp $(ls $O/r05ai_stats/*/*kernel_stats.csv | head -1) $O/r05ai_train_bf16_8x520_wgrad_lanes_kernel_stats.csv
rm -rf $O/r05ai_stats; tail -2 $O/r05ai_stats.log
