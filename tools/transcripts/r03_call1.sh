#!/bin/bash
# round 3, GPU call 1: the SGPR-wave-index epilogue anomaly - hardware probe + kernel-level characterisation + bench fingerprints
O=gpurun_out/r03a; mkdir -p $O
V=tools/_bin/variants
timeout 300 tools/_bin/hazard_probe > $O/hazard_probe.txt 2>&1
timeout 600 python tools/diag_sgpr_epilogue.py --save /tmp/ffn1_ref.pt > $O/diag_stock.json 2> $O/diag_stock.err
for v in sgpr sgpr_asm_none sgpr_nopcarry sgpr_nopstore sgpr_mul24; do
  SOME_AMD_LIBRARY=$V/$v/libsome_amd.so timeout 600 python tools/diag_sgpr_epilogue.py --ref /tmp/ffn1_ref.pt > $O/diag_$v.json 2> $O/diag_$v.err
done
tools/exp_ab.sh r03a/fp "stock|X=1" "sgpr|SOME_AMD_LIBRARY=$V/sgpr/libsome_amd.so" "nopcarry|SOME_AMD_LIBRARY=$V/sgpr_nopcarry/libsome_amd.so" \
  "nopstore|SOME_AMD_LIBRARY=$V/sgpr_nopstore/libsome_amd.so" "mul24|SOME_AMD_LIBRARY=$V/sgpr_mul24/libsome_amd.so" \
  "sgpr_b|SOME_AMD_LIBRARY=$V/sgpr/libsome_amd.so" "nopcarry_b|SOME_AMD_LIBRARY=$V/sgpr_nopcarry/libsome_amd.so" "mul24_b|SOME_AMD_LIBRARY=$V/sgpr_mul24/libsome_amd.so" > $O/fingerprints.txt 2>&1
cat $O/hazard_probe.txt | head -70
cat $O/fingerprints.txt
for f in $O/diag_*.json; do echo $f; cut -c1-1500 $f; done
