#!/bin/bash
# round 5: the generated METH_FASTCALL binding under the training operators - parity (the three training test files, which call through it)
# and step-time A/B against ctypes at the reference's batch shape
O=gpurun_out; mkdir -p $O
export TMPDIR=/tmp
( timeout 500 python -m pytest tests/test_gpu_train_step.py tests/test_gpu_train_ffn16.py tests/test_gpu_train_ops.py -x -q -m gpu 2>&1 | tail -15 ) > $O/r05ak_pytest_fastcall.txt
tail -3 $O/r05ak_pytest_fastcall.txt
for rep in 1 2 3; do
for spec in "ctypes|SOME_AMD_FASTCALL=0" "fastcall|SOME_AMD_FASTCALL=1" "fastcall,in-order-wgrad|SOME_AMD_FASTCALL=1 SOME_AMD_TRAIN_WGRAD_LANES=0"; do
  name=${spec%%|*}; envs=${spec#*|}
  echo -n "$name frames=520: "
  env $envs timeout 300 python tools/train_bench.py --mixed --frames 520 --steps 40 --warmup 8 --digest 2>&1 | tail -2 | tr '\n' ' ' | sed 's/two_head_model lay 3 (mixed bf16): //; s/digest: batch [0-9a-f ]*| grad per step.*| parameters/| parameters/'
  echo
done; done | tee $O/r05ak_train_ab.txt
