# round-4 side evidence: batch-size sweep of the inference step, rocprofv3 kernel stats of the two-lane bf16 training step at two shapes
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
for B in 1 2 4 8 16 32 64; do
  python bench.py --batch $B --steps 20 --warmup 5 --no-cpu-baseline --no-latency --no-f32-leg --no-secondary --no-kernel-profile --no-live-pmc 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B=$B clips of 30 s: %.3f ms per step, %.0f audio-s/s' % (d['ms_per_step'], d['value']))"
done > $O/r04s_batch_size_sweep.txt
cat $O/r04s_batch_size_sweep.txt
for f in 520 2584; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/r04s_stats_$f -- python tools/train_bench.py --frames $f --mixed --steps 20 --warmup 5 > $O/r04s_train_bench_$f.txt 2>&1
  cp $(ls $O/r04s_stats_$f/*/*kernel_stats.csv | head -1) $O/r04s_train_bf16_8x${f}_kernel_stats.csv
  rm -rf $O/r04s_stats_$f
  tail -1 $O/r04s_train_bench_$f.txt
done
