#!/bin/bash
# round 3, GPU call 13: XCD column-split tile map (half of W per L2) - per-shape timing, step A/B, note fingerprint
O=gpurun_out/r03p; mkdir -p $O
for f in 1 3 7; do
  echo "== SOME_AMD_GEMM_FLAGS=$f" >> $O/gemm_bench.txt
  SOME_AMD_GEMM_FLAGS=$f python tools/gemm_bench.py --iters 30 2>&1 | grep -v amdgpu.ids >> $O/gemm_bench.txt
done
for f in 1 3 7 1 3; do
  SOME_AMD_GEMM_FLAGS=$f python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-f32-leg --no-secondary --no-live-pmc --no-latency > $O/bench_flags$f.json 2> $O/bench_flags$f.err
  python - <<PY
import json
d = json.loads(open('$O/bench_flags$f.json').read().strip().split('\n')[-1])
print('flags $f', d['value'], d['ms_per_step'], d.get('notes_decoded_last_step'), [ (k['name'][:24], k['avg_ms']) for k in d['kernels'][:8] ])
PY
done
cat $O/gemm_bench.txt
