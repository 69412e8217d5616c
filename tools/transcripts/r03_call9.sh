#!/bin/bash
# round 3, GPU call 9: 16-bit-storage FFN path - kernel tests, reference-step gates, step time A/B (SOME_AMD_TRAIN_FFN16)
O=gpurun_out/r03k; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_train_ffn16.py -x -q 2>&1 | tail -15 > $O/pytest_ffn16.txt
timeout 900 python -m pytest tests/test_gpu_train_step.py tests/test_gpu_train_ops.py -x -q 2>&1 | tail -15 > $O/pytest_train.txt
for v in 1 0; do
  SOME_AMD_TRAIN_FFN16=$v python tools/train_bench.py --mixed --operand bf16 --steps 10 --warmup 3 2>&1 | grep -v amdgpu.ids > $O/train_bench_ffn16_$v.txt
done
SOME_AMD_TRAIN_FFN16=1 python tools/train_bench.py --mixed --operand bf16 --frames 10000 --steps 4 --warmup 2 2>&1 | grep -v amdgpu.ids >> $O/train_bench_ffn16_1.txt
SOME_AMD_TRAIN_FFN16=0 python tools/train_bench.py --mixed --operand bf16 --frames 10000 --steps 4 --warmup 2 2>&1 | grep -v amdgpu.ids >> $O/train_bench_ffn16_0.txt
tail -n 20 $O/*.txt
