#!/bin/bash
# round 3, GPU call 18: training GPU tests + the epoch at the reference's batch shape, tape on / off (3 h dataset = BASELINE configs[4])
O=gpurun_out/r03t; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_train_step.py tests/test_gpu_train_ops.py tests/test_gpu_train_ffn16.py tests/test_gpu_multiprocess.py -x -q 2>&1 | tail -6 > $O/pytest_train.txt
python tools/make_train_dataset.py --dir /tmp/some_ds --hours 3 > $O/make_ds.txt 2>&1
for r in 1 2; do for v in 0 1; do
  echo "TAPE=$v" >> $O/epoch.txt
  SOME_AMD_TRAIN_TAPE=$v python tools/train_epoch_bench.py --dir /tmp/some_ds 2>&1 | grep -v amdgpu.ids >> $O/epoch.txt
done; done
cat $O/pytest_train.txt; python - <<'PY'
import json
for l in open('gpurun_out/r03t/epoch.txt'):
    if l.startswith('TAPE'): print(l.strip()); continue
    d = json.loads(l); print('  ', d['epoch_wall_s'], d['audio_s_per_s_trained'], d['step_ms'], d['host_enqueue_ms_mean'], d['step_ms_first_quarter_mean'], d['step_ms_last_quarter_mean'], d['device_allocs_during_epoch'])
PY
