#!/bin/bash
# round 5: the placed attention kernel - correctness gates, then interleaved A/B bench legs (SOME_AMD_ATTN_V1=1 = the round-4 kernel)
O=gpurun_out; mkdir -p $O; TAG=${1:-r05c}
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity.py -x -q -k "attention or varlen or full_size or packing" 2>&1 | tail -15 > $O/${TAG}_pytest_attn.txt
tail -6 $O/${TAG}_pytest_attn.txt
tools/exp_ab.sh $TAG "v1a|SOME_AMD_ATTN_V1=1" "placed_a|SOME_AMD_ATTN_V1=0" "v1b|SOME_AMD_ATTN_V1=1" "placed_b|SOME_AMD_ATTN_V1=0"
