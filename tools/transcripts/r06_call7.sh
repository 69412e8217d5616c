#!/bin/bash
# round 6, call 7: the opt-in fast attention mode (gates + step A/B), hardware counters of the persistent stream GEMM against the default
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out
python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -s -k "fast_mode or qkv_attention" 2>&1 | grep -E "fast =|passed|failed|Error|assert" | tail -30 > $O/r06g_pytest_fast_kernels.txt
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -s -k "fullsize_batch and fast" 2>&1 | grep -E "full_|passed|failed|Error|assert" | tail -12 > $O/r06g_pytest_fast_fullsize.txt
bash tools/exp_ab.sh r06g "base1|SOME_AMD_PRECISION=f16x3" "fast1|SOME_AMD_PRECISION=f16x3_fast" "fast2x|SOME_AMD_PRECISION=f16x3_fast SOME_AMD_ATTN_FAST=2" "base2|SOME_AMD_PRECISION=f16x3" "fast1b|SOME_AMD_PRECISION=f16x3_fast" > $O/r06g_step_ab.txt 2>&1
for f in 1 3; do
  ( cd /tmp && SOME_AMD_GEMM_FLAGS=$f rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc_flags$f -- python $GRAFT_REPO_ROOT/tools/gemm_bench.py --tile 2 --iters 60 ) > $O/r06g_pmc_flags$f.log 2>&1
  python tools/pmc_summary.py /tmp/pmc_flags$f > $O/r06g_pmc_gemm_flags$f.json 2>> $O/r06g_pmc_flags$f.log
done
cat $O/r06g_pytest_fast_kernels.txt $O/r06g_pytest_fast_fullsize.txt $O/r06g_step_ab.txt
python - <<'PY'
import json
for f in (1, 3):
    d = json.load(open(f'gpurun_out/r06g_pmc_gemm_flags{f}.json'))
    for k, v in d.items():
        if 'MfmaUtil_percent' in v and 'hgemm3' in k:
            print(f, k[:60], round(v['MfmaUtil_percent'], 1), round(v['effective_clock_MHz']), v['_duration_ns']['dispatches'], round(v['_duration_ns']['mean_per_dispatch'] / 1e3, 1))
PY
