#!/bin/bash
# round 3, GPU call 12: deep-ring small-M GEMM (tile 5) - kernel tests, per-shape timing at 2584 rows, single-clip latency A/B
O=gpurun_out/r03n; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "gemm_f16x3" 2>&1 | tail -5 > $O/pytest_kernels.txt
for t in 4 5; do python tools/gemm_bench.py --M 2584 --tile $t --iters 50 2>&1 | grep -v amdgpu.ids > $O/gemm_bench_tile$t.txt; done
for t in -1 5; do
  SOME_AMD_TILE=$t python bench.py --batch 1 --steps 20 --warmup 5 --no-cpu-baseline --no-f32-leg --no-secondary --no-live-pmc > $O/bench_b1_tile$t.json 2> $O/bench_b1_tile$t.err
done
tail -n 8 $O/*.txt
python - <<'PY'
import json
for t in (-1, 5):
    d = json.loads(open(f'gpurun_out/r03n/bench_b1_tile{t}.json').read().strip().split('\n')[-1])
    print(t, d['ms_per_step'], d.get('p50_clip_latency_ms'), d.get('notes_decoded_last_step'))
    for k in d.get('kernels', [])[:8]:
        print('   ', k['name'], k['avg_ms'])
PY
