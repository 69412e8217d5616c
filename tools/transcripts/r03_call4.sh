#!/bin/bash
# round 3, GPU call 4: MX-FP6 cross terms in the attention P V product - kernel tests, parity, A/B timing
O=gpurun_out/r03e; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "qkv_attention" > $O/pytest_kernels.txt 2>&1; tail -5 $O/pytest_kernels.txt
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -s -k "fullsize_batch or model_forward_golden or varlen or large_ragged or twenty" > $O/pytest_parity.txt 2>&1; tail -4 $O/pytest_parity.txt; grep "32 x 30 s" $O/pytest_parity.txt
tools/exp_ab.sh r03e/ab "mx0|SOME_AMD_ATTN_MX=0" "mx1|SOME_AMD_ATTN_MX=1" "mx0_b|SOME_AMD_ATTN_MX=0" "mx1_b|SOME_AMD_ATTN_MX=1" 2>&1 | cut -c1-420
