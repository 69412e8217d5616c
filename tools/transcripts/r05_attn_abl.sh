#!/bin/bash
# round 5: ablation builds of the placed attention kernel (SOME_ATTN_ABL mask: 1 no softmax VALU, 2 no fragment reads, 4 no staging,
# 8 no barrier) - attention kernel time only (results are garbage by construction)
O=gpurun_out; mkdir -p $O; TAG=${1:-r05e}
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-latency --no-f32-leg --no-secondary --no-e2e --no-train --no-live-pmc"
for v in base "$@"; do
  [ "$v" = "$TAG" ] && continue
  if [ "$v" = base ]; then L=""; else L="SOME_AMD_LIBRARY=tools/_bin/variants/$v/libsome_amd.so"; fi
  env $L timeout 200 $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
ks={k['name']:k['avg_ms'] for k in d.get('kernels',[])}
print('$v', 'attention_ms', ks.get('attention'), 'step_ms', d['ms_per_step'])
"
done | tee $O/${TAG}_attn_ablation.txt
