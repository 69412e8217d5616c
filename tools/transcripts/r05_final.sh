#!/bin/bash
# round 5, closing evidence on one box: the GPU suite, smoke(), the default bench line + rocprofv3 kernel stats + PMC passes
# (tools/collect_profiles.sh), kernel stats of the training step in its default configuration
O=gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
( timeout 420 python -m pytest tests -x -q -m gpu 2>&1 | tail -12 ) > $O/r05zy_pytest_gpu_tail.txt
tail -2 $O/r05zy_pytest_gpu_tail.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee $O/r05zy_smoke.txt
bash tools/collect_profiles.sh r05zy > $O/r05zy_collect.log 2>&1
tail -c 600 $O/r05zy_bench.json
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r05zy_tstats -- python tools/train_bench.py --mixed --frames 520 --steps 25 --warmup 5 > $O/r05zy_train_stats.log 2>&1
cp $(ls $O/r05zy_tstats/*/*kernel_stats.csv | head -1) $O/r05zy_train_bf16_8x520_kernel_stats.csv; rm -rf $O/r05zy_tstats
tail -1 $O/r05zy_train_stats.log | cut -c1-200
