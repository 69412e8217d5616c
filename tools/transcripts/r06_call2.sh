#!/bin/bash
# round 6, call 2: the single-stage two-workgroups-per-CU GEMM tile (SOME_AMD_TILE=5): kernel gates, per-shape A/B, step A/B
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out
python -m pytest tests/test_gpu_kernels.py -x -q -m gpu 2>&1 | tail -8 > $O/r06b_pytest_kernels.txt
python -m pytest tests/test_gpu_multiprocess.py -x -q -m gpu -s -k "eight_rank or eight_gloo_ranks and train" 2>&1 | grep -E "8 ranks vs|passed|failed|Error" | tail -8 > $O/r06b_pytest_w8.txt
for rep in 1 2; do
  for t in 2 5; do echo "== tile $t (rep $rep)"; python tools/gemm_bench.py --tile $t --iters 30; done
done > $O/r06b_gemm_bench.txt 2>&1
bash tools/exp_ab.sh r06b "base1|SOME_AMD_TILE=-1" "single1|SOME_AMD_TILE=5" "base2|SOME_AMD_TILE=-1" "single2|SOME_AMD_TILE=5" > $O/r06b_step_ab.txt 2>&1
cat $O/r06b_pytest_kernels.txt $O/r06b_pytest_w8.txt $O/r06b_gemm_bench.txt $O/r06b_step_ab.txt
