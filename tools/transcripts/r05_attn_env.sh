#!/bin/bash
# round 5: attention kernels selected by environment switches through the bench's per-kernel table, interleaved, two repetitions
O=gpurun_out; mkdir -p $O; TAG=$1; shift
B="python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-latency --no-f32-leg --no-secondary --no-e2e --no-train --no-live-pmc"
for rep in 1 2; do
for spec in "$@"; do
  name=${spec%%|*}; envs=${spec#*|}
  env $envs timeout 200 $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
ks={k['name']:k['avg_ms'] for k in d.get('kernels',[])}
print('$name', 'attention_ms', ks.get('attention'), 'step_ms', d['ms_per_step'], 'notes', d.get('notes_decoded_last_step'))
"
done; done | tee $O/${TAG}_attn_env.txt
