#!/bin/bash
# round 3, GPU call 20: end state of the training path - kernel stats at 8 x 2584 (bf16), the epoch of BASELINE configs[4] at its own size
O=gpurun_out/r03y; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python tools/train_bench.py --mixed --operand bf16 --steps 5 --warmup 2 > $O/train_bench_under_rocprof.txt 2> $O/stats.log
cp $(ls $O/stats/*/*kernel_stats.csv | head -1) $O/train_bf16_kernel_stats.csv
rm -rf $O/stats
python tools/make_train_dataset.py --dir /tmp/some_ds --hours 3 > $O/make_ds.txt 2>&1
for r in 1 2; do python tools/train_epoch_bench.py --dir /tmp/some_ds 2>&1 | grep -v amdgpu.ids >> $O/epoch.txt; done
for f in 2584 10000 520; do python tools/train_bench.py --mixed --operand bf16 --frames $f --steps 8 --warmup 3 2>&1 | grep -v amdgpu >> $O/train_bench.txt; done
python tools/train_bench.py --steps 5 --warmup 2 2>&1 | grep -v amdgpu >> $O/train_bench.txt
cat $O/train_bench.txt; cat $O/epoch.txt | cut -c1-900
