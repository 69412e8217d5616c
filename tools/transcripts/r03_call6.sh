#!/bin/bash
# round 3, GPU call 6: where does the small-batch training step go (GPU busy vs wall)?
O=gpurun_out/r03g; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python tools/make_train_dataset.py --dir /tmp/ds05 --hours 0.5 > $O/make_ds.txt 2>&1; tail -2 $O/make_ds.txt
python tools/train_epoch_bench.py --dir /tmp/ds05 > $O/epoch_plain.json 2> $O/epoch_plain.err; cat $O/epoch_plain.json | cut -c1-900
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python tools/train_epoch_bench.py --dir /tmp/ds05 > $O/epoch_prof.json 2> $O/prof.log
cp $(ls $O/prof/*/*kernel_stats.csv | head -1) $O/train_epoch_kernel_stats.csv; rm -rf $O/prof
python - <<'PY'
import csv, json
rows = list(csv.DictReader(open('gpurun_out/r03g/train_epoch_kernel_stats.csv')))
tot = sum(float(r['TotalDurationNs']) for r in rows); calls = sum(int(r['Calls']) for r in rows)
d = json.loads(open('gpurun_out/r03g/epoch_prof.json').read().strip().splitlines()[-1])
steps = d['updates'] + 2
print('kernel time total %.1f ms over %d launches; per update: %.2f ms GPU busy, %d launches; wall per update %.2f ms' % (tot / 1e6, calls, tot / 1e6 / steps, calls // steps, d['step_ms']['mean']))
for r in sorted(rows, key=lambda r: -float(r['TotalDurationNs']))[:14]:
    print('  %-70s calls %6s  total %8.2f ms  avg %7.1f us' % (r['Name'][:70], r['Calls'], float(r['TotalDurationNs']) / 1e6, float(r['AverageNs']) / 1e3))
PY
