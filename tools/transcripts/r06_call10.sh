#!/bin/bash
# round 6, call 10: DDP overlap model with a HIGH-PRIORITY communication stream; corrected GEMM workgroup timeline
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out
for spec in "default_prio|" "lanes_only_prio|SOME_AMD_TRAIN_WGRAD_LANES=0" "one_stream_prio|SOME_AMD_TRAIN_LANES=1 SOME_AMD_TRAIN_WGRAD_LANES=0"; do
  name=${spec%%|*}; envs=${spec#*|}
  echo "=== $name ($envs) --comm-priority -1"
  env $envs python tools/ddp_overlap_bench.py --buckets 8 16 32 64 --steps 20 --comm-priority -1 2>&1 | grep -v amdgpu.ids
done > $O/r06j_ddp_overlap_priority.txt 2>&1
tools/_bin/gemm_probe > $O/r06j_gemm_timeline.txt 2>&1
cat $O/r06j_ddp_overlap_priority.txt; grep -v "k-block lengths" $O/r06j_gemm_timeline.txt | cut -c1-400
