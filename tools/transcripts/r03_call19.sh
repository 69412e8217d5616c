#!/bin/bash
# round 3, GPU call 19: kernel stats of the bf16 training step at the reference's batch shape (8 x 520 frames) with the tape
O=gpurun_out/r03u; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python tools/train_bench.py --mixed --operand bf16 --frames 520 --steps 20 --warmup 5 > $O/train_bench_under_rocprof.txt 2> $O/stats.log
cp $(ls $O/stats/*/*kernel_stats.csv | head -1) $O/train_bf16_8x520_kernel_stats.csv
rm -rf $O/stats
cat $O/train_bench_under_rocprof.txt | grep -v amdgpu
