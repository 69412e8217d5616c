#!/bin/bash
# round 6, call 4: persistent stream GEMM - gates, workgroup timelines of the three variants, per-shape and step A/B; B = 1 latency timeline
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out
python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "persistent" 2>&1 | tail -12 > $O/r06d_pytest_persistent.txt
tools/_bin/gemm_probe > $O/r06d_gemm_timeline.txt 2>&1
bash tools/exp_ab.sh r06d "base1|SOME_AMD_GEMM_FLAGS=1" "persist1|SOME_AMD_GEMM_FLAGS=3" "base2|SOME_AMD_GEMM_FLAGS=1" "persist2|SOME_AMD_GEMM_FLAGS=3" > $O/r06d_step_ab.txt 2>&1
( cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/lt_eager -- python $GRAFT_REPO_ROOT/tools/latency_timeline.py run --steps 12 ) > $O/r06d_latency_eager_prof.txt 2>&1
python tools/latency_timeline.py analyse /tmp/lt_eager > $O/r06d_latency_timeline_eager.txt 2>&1
cp /tmp/lt_eager/*/*kernel_trace.csv $O/r06d_latency_kernel_trace.csv 2>/dev/null
cat $O/r06d_pytest_persistent.txt $O/r06d_gemm_timeline.txt $O/r06d_step_ab.txt $O/r06d_latency_timeline_eager.txt
