#!/bin/bash
# round 6, call 12: training epoch A/B (release marks of the weight-gradient operands on / off), LayerNorm 16-byte SPLIT32 stores A/B
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out
DS=/tmp/some_amd_bench/train_3h
python tools/make_train_dataset.py --dir $DS --hours 3 > /dev/null 2>&1
for rep in 1 2; do
  for spec in "marks16|SOME_AMD_TRAIN_WG_MARK_EVERY=16" "nomarks|SOME_AMD_TRAIN_WG_MARK_EVERY=1000000000" "marks64|SOME_AMD_TRAIN_WG_MARK_EVERY=64"; do
    name=${spec%%|*}; envs=${spec#*|}
    echo "== $name rep $rep"; env $envs python tools/train_epoch_bench.py --dir $DS 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['audio_s_per_s_trained'], d['step_ms'], 'enqueue', d['host_enqueue_ms_mean'])"
  done
done > $O/r06k_train_epoch_marks_ab.txt 2>&1
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_determinism.py -x -q -m gpu -k "layernorm or LayerNorm or determin or bench_step" 2>&1 | tail -4 > $O/r06k_pytest_ln_base.txt
SOME_AMD_LIBRARY=tools/_bin/variants/ln16/libsome_amd.so python -m pytest tests/test_gpu_kernels.py tests/test_gpu_determinism.py tests/test_gpu_parity.py -x -q -m gpu -k "layernorm or LayerNorm or determin or fullsize_batch or varlen" 2>&1 | tail -4 > $O/r06k_pytest_ln16.txt
bash tools/exp_ab.sh r06k "base1|SOME_AMD_TILE=-1" "ln16a|SOME_AMD_LIBRARY=tools/_bin/variants/ln16/libsome_amd.so" "base2|SOME_AMD_TILE=-1" "ln16b|SOME_AMD_LIBRARY=tools/_bin/variants/ln16/libsome_amd.so" > $O/r06k_step_ab_ln16.txt 2>&1
cat $O/r06k_train_epoch_marks_ab.txt $O/r06k_pytest_ln_base.txt $O/r06k_pytest_ln16.txt; cut -c1-420 $O/r06k_step_ab_ln16.txt
