#!/bin/bash
# round 3, GPU call 10: kernel stats of the bf16 training step with the 16-bit FFN path
O=gpurun_out/r03l; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python tools/train_bench.py --mixed --operand bf16 --steps 5 --warmup 2 > $O/train_bench_under_rocprof.txt 2> $O/stats.log
cp $(ls $O/stats/*/*kernel_stats.csv | head -1) $O/train_bf16_ffn16_kernel_stats.csv
rm -rf $O/stats
head -30 $O/train_bf16_ffn16_kernel_stats.csv | cut -c1-200
