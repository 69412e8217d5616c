#!/bin/bash
# round 6, call 9: DDP overlap model under different stream / hardware-queue settings; training tests after the ops.py split; bench legs
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out
for spec in "default|" "hwq8|GPU_MAX_HW_QUEUES=8" "one_stream|SOME_AMD_TRAIN_LANES=1 SOME_AMD_TRAIN_WGRAD_LANES=0" "lanes_only|SOME_AMD_TRAIN_WGRAD_LANES=0" "lanes_only_hwq8|SOME_AMD_TRAIN_WGRAD_LANES=0 GPU_MAX_HW_QUEUES=8"; do
  name=${spec%%|*}; envs=${spec#*|}
  echo "=== $name ($envs)"
  env $envs python tools/ddp_overlap_bench.py --buckets 16 32 64 --steps 20 2>&1 | grep -v amdgpu.ids
done > $O/r06i_ddp_overlap_streams.txt 2>&1
python -m pytest tests/test_gpu_train_step.py tests/test_gpu_train_ffn16.py tests/test_gpu_train_ops.py -x -q -m gpu 2>&1 | tail -6 > $O/r06i_pytest_train.txt
cat $O/r06i_ddp_overlap_streams.txt $O/r06i_pytest_train.txt
