#!/bin/bash
# round 3, GPU call 11: tile / ring-shape variants of the 16-bit-storage GEMM (fp32 epilogue), per layer shape
for v in 0 2 3 4 5; do
  echo "== variant $v"
  SOME_AMD_G16S_VARIANT=$v python tools/train_gemm_bench.py --stored16 2>&1 | grep -E "fwd|dgrad" | grep -v epilogue
done
