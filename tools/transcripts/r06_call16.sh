#!/bin/bash
# round 6, call 16: FFN1's SPLIT32 epilogue with whole-line stores through an LDS patch (SOME_AMD_GEMM_FLAGS bit 2): gate + A/B
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out
python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "persistent" 2>&1 | tail -6 > $O/r06n_pytest_lines.txt
bash tools/exp_ab.sh r06n "base1|SOME_AMD_GEMM_FLAGS=3" "lines1|SOME_AMD_GEMM_FLAGS=7" "base2|SOME_AMD_GEMM_FLAGS=3" "lines2|SOME_AMD_GEMM_FLAGS=7" > $O/r06n_step_ab_lines.txt 2>&1
cat $O/r06n_pytest_lines.txt; cut -c1-260 $O/r06n_step_ab_lines.txt
