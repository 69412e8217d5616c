#!/bin/bash
# round 3, GPU call 15: ablation of the big-batch split-f16 GEMM k-loop (variant libraries, wrong results by construction)
O=gpurun_out/r03q; mkdir -p $O
for v in stock abl_noepi abl_mfmaonly abl_noloop abl_noepi_nomfma; do
  echo "== $v" >> $O/gemm_ablation.txt
  if [ $v = stock ]; then python tools/gemm_bench.py --iters 30 2>&1 | grep -v amdgpu.ids >> $O/gemm_ablation.txt
  else SOME_AMD_LIBRARY=tools/_bin/variants/$v/libsome_amd.so python tools/gemm_bench.py --iters 30 2>&1 | grep -v amdgpu.ids >> $O/gemm_ablation.txt; fi
done
cat $O/gemm_ablation.txt
