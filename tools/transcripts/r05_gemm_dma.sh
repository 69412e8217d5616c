#!/bin/bash
# round 5: split-f16 GEMMs with LDS-DMA staging (apply tools/patches/r05_gemm_sign_flip_and_lds_dma.patch, then
#   python tools/build_variant.py gdma gemm_f16x3.hip -DSOME_GEMM_DMA=1) - correctness gates, then interleaved A/B
O=gpurun_out; mkdir -p $O; TAG=${1:-r05y}
V=tools/_bin/variants/gdma/libsome_amd.so
SOME_AMD_LIBRARY=$V timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity.py -x -q -k "gemm or forward or varlen or fullsize or packing" 2>&1 | tail -4
B="python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-latency --no-f32-leg --no-secondary --no-e2e --no-train --no-live-pmc"
for rep in 1 2; do for v in stock gdma; do
  if [ $v = stock ]; then L="X=1"; else L="SOME_AMD_LIBRARY=$V"; fi
  env $L timeout 200 $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
ks={k['name']:k['avg_ms'] for k in d.get('kernels',[])}
print('$v', 'step_ms', d['ms_per_step'], 'notes', d.get('notes_decoded_last_step'), ' '.join(f'{n}={v:.4f}' for n,v in ks.items() if n.startswith('gemm')))
"
done; done | tee $O/${TAG}_gemm_dma.txt
