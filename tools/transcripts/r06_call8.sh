#!/bin/bash
# round 6, call 8: the whole GPU suite with the round's defaults (persistent stream GEMM on), the DDP overlap model, the measured-only
# FAST = 2 attention variant at model level
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out
python -m pytest tests -m gpu -q -x 2>&1 | tail -15 > $O/r06h_pytest_gpu_tail.txt
python tools/ddp_overlap_bench.py > $O/r06h_ddp_overlap.txt 2>&1
SOME_AMD_ATTN_FAST=2 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -s -k "fullsize_batch and fast" 2>&1 | grep -E "full_|passed|failed|Error|assert" | tail -12 > $O/r06h_pytest_fast2_fullsize.txt
cat $O/r06h_pytest_gpu_tail.txt $O/r06h_ddp_overlap.txt $O/r06h_pytest_fast2_fullsize.txt
