#!/bin/bash
# A/B experiment runner on the MI355X box (one gpurun call): bench legs with library variants / run-time flags, interleaved,
# each leg one JSON line under gpurun_out/<tag>_<name>.json.  Usage: tools/exp_ab.sh <tag> "name|ENV=.. ENV=.." ...
TAG=$1; shift
O=gpurun_out; mkdir -p $O
B="python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-latency --no-f32-leg --no-fast-leg --no-calibration --no-secondary --no-e2e --no-train --no-live-pmc"
for spec in "$@"; do
    name=${spec%%|*}; envs=${spec#*|}
    env $envs timeout 300 $B > $O/${TAG}_${name}.json 2> $O/${TAG}_${name}.err
    python - "$O/${TAG}_${name}.json" "$name" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    ks = {k['name']: k['avg_ms'] for k in d.get('kernels', [])}
    print(sys.argv[2], 'ms/step', d['ms_per_step'], 'notes', d.get('notes_decoded_last_step'),
          ' '.join(f"{n.split('[')[0][:14]}{('[' + n.split('[')[1]) if '[' in n else ''}={v:.4f}" for n, v in ks.items() if v > 0.1))
except Exception as e:
    print(sys.argv[2], 'FAILED', e)
PY
done
