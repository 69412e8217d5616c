#!/bin/bash
# round 3, GPU call 16: kernel stats of the bf16 training step at 8 x 10 000 frames
O=gpurun_out/r03r; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python tools/train_bench.py --mixed --operand bf16 --frames 10000 --steps 3 --warmup 1 > $O/train_bench_under_rocprof.txt 2> $O/stats.log
cp $(ls $O/stats/*/*kernel_stats.csv | head -1) $O/train_bf16_10000_kernel_stats.csv
rm -rf $O/stats
