#!/bin/bash
# round 6, call 13: four-channel depthwise kernel (gates + A/B), LayerNorm row-count / non-temporal variants, grouped vs dual-stream with the persistent GEMMs
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_determinism.py tests/test_gpu_parity.py -x -q -m gpu -k "dwconv or determin or fullsize_batch or varlen" 2>&1 | tail -4 > $O/r06l_pytest_dwconv4.txt
bash tools/exp_ab.sh r06l "dw1a|SOME_AMD_DWCONV4=0" "dw4a|SOME_AMD_DWCONV4=1" "dw1b|SOME_AMD_DWCONV4=0" "dw4b|SOME_AMD_DWCONV4=1" "ln_r3|SOME_AMD_LIBRARY=tools/_bin/variants/ln16r3/libsome_amd.so" "ln_r4|SOME_AMD_LIBRARY=tools/_bin/variants/ln16r4/libsome_amd.so" "ln_nt|SOME_AMD_LIBRARY=tools/_bin/variants/ln16nt/libsome_amd.so" "base|SOME_AMD_TILE=-1" > $O/r06l_step_ab.txt 2>&1
B="python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-latency --no-f32-leg --no-fast-leg --no-calibration --no-secondary --no-e2e --no-train --no-live-pmc --no-kernel-profile"
for rep in 1 2; do for m in 1 0; do echo "dual_stream=$m rep $rep: $(SOME_AMD_DUAL_STREAM=$m $B 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")"; done; done > $O/r06l_dual_vs_grouped.txt 2>&1
cat $O/r06l_pytest_dwconv4.txt; cut -c1-330 $O/r06l_step_ab.txt; cat $O/r06l_dual_vs_grouped.txt
