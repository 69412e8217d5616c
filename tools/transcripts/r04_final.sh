# round-4 closing evidence (one gpurun call): full GPU suite, the default bench line, a learning-curve run of the two-lane / async trainer
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -12 > $O/r04z_pytest_gpu_tail.txt
tail -3 $O/r04z_pytest_gpu_tail.txt
python bench.py > $O/r04z_bench.json 2> $O/r04z_bench.err
tail -c 400 $O/r04z_bench.json
python - <<'PY'
import yaml, sys
sys.path.insert(0, '.')
from some_amd.configs import get_config
cfg = get_config('two_head_model', lay=2)
cfg['pl_trainer_precision'] = 'bf16'
cfg['lr_scheduler_args'] = {'scheduler_cls': 'lr_scheduler.scheduler.WarmupLR', 'warmup_steps': 40, 'min_lr': 1e-5}
cfg['optimizer_args'] = dict(cfg.get('optimizer_args', {}), lr=4e-4)
cfg['val_check_interval'] = 150
cfg['log_interval'] = 50
yaml.safe_dump(cfg, open('/tmp/r04_curve.yaml', 'w'))
PY
S=$(date +%s.%N)
python train.py --config /tmp/r04_curve.yaml --exp_name q --work_dir /tmp/r04_curve --synthetic 96 --max_updates 600 --log_interval 100 > $O/r04z_train_learning_curve.txt 2>&1
echo "wall $(python -c "import time; print(round(time.time() - $S, 1))") s" >> $O/r04z_train_learning_curve.txt
grep -E "validation|step 600|wall" $O/r04z_train_learning_curve.txt | tail -6
for f in 520 2584 10000; do python tools/train_bench.py --frames $f --mixed --steps 10 --warmup 3 2>&1 | tail -1; done > $O/r04z_train_bench.txt; cat $O/r04z_train_bench.txt
