#!/bin/bash
# round 5: device-side attention operand preparation in the training step - tests, eager-operator census, A / B timing
O=gpurun_out; mkdir -p $O
python -m pytest tests/test_gpu_train_ffn16.py tests/test_gpu_train_step.py -x -q -m gpu 2>&1 | tail -5 > $O/r05ah_pytest.txt
cat $O/r05ah_pytest.txt
python tools/train_eager_ops.py > $O/r05ah_eager_ops.txt 2>&1
head -50 $O/r05ah_eager_ops.txt
for i in 1 2; do
  for v in 0 1; do
    echo -n "device_prep=$v " >> $O/r05ah_train_ab.txt
    SOME_AMD_TRAIN_DEVICE_PREP=$v python tools/train_bench.py --frames 520 --mixed --steps 50 --warmup 10 2>/dev/null | tail -1 >> $O/r05ah_train_ab.txt
  done
done
cat $O/r05ah_train_ab.txt
