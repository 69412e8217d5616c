#!/bin/bash
# round 5: binding (ctypes / generated fastcall) x weight-gradient lanes (in order / side streams), step time at the reference's batch shape
O=gpurun_out; mkdir -p $O
for rep in 1 2 3; do
for spec in "ctypes,in-order|SOME_AMD_FASTCALL=0 SOME_AMD_TRAIN_WGRAD_LANES=0" "ctypes,lanes|SOME_AMD_FASTCALL=0 SOME_AMD_TRAIN_WGRAD_LANES=1" "fastcall,in-order|SOME_AMD_FASTCALL=1 SOME_AMD_TRAIN_WGRAD_LANES=0" "fastcall,lanes|SOME_AMD_FASTCALL=1 SOME_AMD_TRAIN_WGRAD_LANES=1"; do
  name=${spec%%|*}; envs=${spec#*|}
  echo -n "$name frames=520: "
  env $envs timeout 200 python tools/train_bench.py --mixed --frames 520 --steps 60 --warmup 10 2>&1 | tail -1 | sed 's/two_head_model lay 3 (mixed bf16): //'
done; done | tee $O/r05al_train_ab.txt
