#!/bin/bash
# round 6, call 14: closing evidence with the final defaults (persistent GEMM, 16-byte LayerNorm stores): GPU suite, smoke, bench + profiles
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out
python -m pytest tests -m gpu -q -x 2>&1 | tail -12 > $O/r06zz_pytest_gpu_tail.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/r06zz_smoke.txt 2>&1
bash tools/collect_profiles.sh r06zz > $O/r06zz_collect.log 2>&1
cat $O/r06zz_pytest_gpu_tail.txt $O/r06zz_smoke.txt; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06zz_bench.json').read().strip().splitlines()[-1])
print({k: d.get(k) for k in ('value','ms_per_step','value_normalised','ms_per_step_normalised','p50_clip_latency_ms','p50_clip_latency_graph_replay_ms')})
print(d['box']['mfma_tf'], d['box']['copy_gbs'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d.get('live_pmc'))
print(d['fast_mode']['ms_per_step'], d['exact_f32_mode']['value'], d['secondary']['value'], d['e2e_batch_infer']['wall_s'], d['train_epoch']['audio_s_per_s_trained'], d['cpu_baseline']['value'])
for k in d['kernels'][:10]: print(k['name'], k['avg_ms'])
PY
