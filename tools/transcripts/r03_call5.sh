#!/bin/bash
# round 3, GPU call 5: fused split-K at small batches (latency), multi-process tests
O=gpurun_out/r03f; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_multiprocess.py -x -q -s > $O/pytest_multiprocess.txt 2>&1; tail -4 $O/pytest_multiprocess.txt; grep "rows identical" $O/pytest_multiprocess.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "varlen or golden or vs_oracle or dual_stream or hip_graph" > $O/pytest_parity.txt 2>&1; tail -3 $O/pytest_parity.txt
L="python bench.py --batch 1 --steps 30 --warmup 5 --no-cpu-baseline --no-f32-leg --no-secondary --no-kernel-profile --no-live-pmc"
for v in "off|SOME_AMD_SPLITK_MAX=1" "s4k8|SOME_AMD_SPLITK_MAX=4" "s4k4|SOME_AMD_SPLITK_MAX=4 SOME_AMD_SPLITK_MINK=4" "s2k8|SOME_AMD_SPLITK_MAX=2" "s3k8|SOME_AMD_SPLITK_MAX=3" "off_b|SOME_AMD_SPLITK_MAX=1" "s4k8_b|SOME_AMD_SPLITK_MAX=4"; do
  name=${v%%|*}; envs=${v#*|}
  env $envs timeout 300 $L > $O/lat_$name.json 2> $O/lat_$name.err
  python - $O/lat_$name.json $name <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], 'ms/step(B=1)', d['ms_per_step'], 'p50', d.get('p50_clip_latency_ms'), 'notes', d['notes_decoded_last_step'])
except Exception as e: print(sys.argv[2], 'FAILED', e)
PY
done
