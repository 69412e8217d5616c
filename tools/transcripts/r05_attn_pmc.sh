#!/bin/bash
# round 5: SQ counters of the attention kernel, round-4 kernel (SOME_AMD_ATTN_V1=1) vs the placed one; two passes each (8 SQ slots per pass)
O=gpurun_out; mkdir -p $O; TAG=${1:-r05d}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
P="python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-profile --no-latency --no-f32-leg --no-secondary --no-e2e --no-train --no-live-pmc"
for v in 1 0; do
  SOME_AMD_ATTN_V1=$v SOME_AMD_DUAL_STREAM=0 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $O/${TAG}_pa$v -- $P > /dev/null 2>&1
  SOME_AMD_ATTN_V1=$v SOME_AMD_DUAL_STREAM=0 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F16 GRBM_GUI_ACTIVE SQ_INSTS_SALU --output-format csv -d $O/${TAG}_pb$v -- $P > /dev/null 2>&1
  python tools/pmc_summary.py $O/${TAG}_pa$v $O/${TAG}_pb$v > $O/${TAG}_pmc_attn_v1_$v.json
  rm -rf $O/${TAG}_pa$v $O/${TAG}_pb$v
  python - $O/${TAG}_pmc_attn_v1_$v.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k, v in d.items():
    if 'attention' in k:
        print(k[:40], {c: (round(x['mean_per_dispatch'] / 1e6, 1) if isinstance(x, dict) else round(x, 1)) for c, x in v.items()})
PY
done
