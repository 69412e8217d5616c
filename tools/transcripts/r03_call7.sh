#!/bin/bash
# round 3, GPU call 7: evidence of the committed state - whole GPU suite, default bench + rocprofv3 stats + PMC passes, e2e / train legs
O=gpurun_out; mkdir -p $O/r03h
timeout 2400 python -m pytest tests -m gpu -x -q > $O/r03h/pytest_gpu.txt 2>&1; tail -4 $O/r03h/pytest_gpu.txt
tools/collect_profiles.sh r03f > $O/r03h/collect.log 2>&1
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03f_bench.json').read().strip().splitlines()[-1])
print('default bench:', d['value'], 'audio-s/s', d['ms_per_step'], 'ms; p50', d.get('p50_clip_latency_ms'), 'f32', d.get('exact_f32_mode',{}).get('value'), 'quant', d.get('secondary',{}).get('value'), 'cpu', d.get('cpu_baseline',{}).get('value'))
print('roofline', {k:d['roofline'].get(k) for k in ('achieved','frac','traffic','traffic_over_algorithmic_bytes','avg_launch_ms')}, 'live', d.get('live_pmc'))
PY
head -12 $O/r03f_kernel_stats.csv | cut -c1-200
timeout 2400 python bench.py --e2e --train --no-cpu-baseline --no-f32-leg --no-secondary > $O/r03h/bench_e2e_train.json 2> $O/r03h/bench_e2e_train.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03h/bench_e2e_train.json').read().strip().splitlines()[-1])
print('e2e', json.dumps(d.get('e2e_batch_infer'))[:1400]); print('train', json.dumps(d.get('train_epoch'))[:900])
PY
