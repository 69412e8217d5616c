#!/bin/bash
# round 3, GPU call 17: the training epoch at the reference's batch shape with / without the 16-bit FFN path
O=gpurun_out/r03s; mkdir -p $O
python tools/make_train_dataset.py --dir /tmp/some_ds --hours 1 > $O/make_ds.txt 2>&1
for r in 1 2; do for v in 0 1; do
  echo "FFN16=$v" >> $O/epoch.txt
  SOME_AMD_TRAIN_FFN16=$v python tools/train_epoch_bench.py --dir /tmp/some_ds 2>&1 | grep -v amdgpu.ids >> $O/epoch.txt
done; done
cat $O/epoch.txt
