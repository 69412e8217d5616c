#!/bin/bash
# round 3, GPU call 3: BASELINE configs[3] / [4] at their own size as bench legs, live PMC, training CLI tests, s_nop 0 variant
O=gpurun_out/r03c; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_train_step.py -x -q -k "cli or binarised or two_rank" > $O/pytest_train_cli.txt 2>&1; tail -3 $O/pytest_train_cli.txt
timeout 600 python tools/diag_sgpr_epilogue.py --save /tmp/ffn1_ref.pt > $O/diag_stock.json 2> $O/diag_stock.err
for v in sgpr sgpr_nopstore0; do
  SOME_AMD_LIBRARY=tools/_bin/variants/$v/libsome_amd.so timeout 600 python tools/diag_sgpr_epilogue.py --ref /tmp/ffn1_ref.pt 2> $O/diag_$v.err | cut -c1-400 > $O/diag_$v.json
done
head -c 300 $O/diag_sgpr.json; echo; head -c 300 $O/diag_sgpr_nopstore0.json; echo
timeout 2400 python bench.py --e2e --train > $O/bench_e2e_train.json 2> $O/bench_e2e_train.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03c/bench_e2e_train.json').read().strip().splitlines()[-1])
print('value', d['value'], 'ms', d['ms_per_step'], 'p50', d.get('p50_clip_latency_ms'))
print('roofline', {k: d['roofline'].get(k) for k in ('achieved','frac','traffic','traffic_over_algorithmic_bytes')})
print('live_pmc', d.get('live_pmc'))
print('e2e', json.dumps(d.get('e2e_batch_infer'))[:1500])
print('train', json.dumps(d.get('train_epoch'))[:1500])
PY
tail -5 $O/bench_e2e_train.err
