#!/bin/bash
# round 6, call 3: GEMM workgroup timeline (tools/gemm_probe.hip), B = 1 latency timeline eager vs graph replay, graph-runner gate
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out
tools/_bin/gemm_probe > $O/r06c_gemm_timeline.txt 2>&1
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "graph" 2>&1 | tail -5 > $O/r06c_pytest_graph.txt
python tools/latency_timeline.py run --steps 14 > $O/r06c_latency_eager.txt 2>&1
python tools/latency_timeline.py run --steps 14 --graph > $O/r06c_latency_graph.txt 2>&1
( cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/lt_eager -- python $GRAFT_REPO_ROOT/tools/latency_timeline.py run --steps 12 ) > $O/r06c_latency_eager_prof.txt 2>&1
python tools/latency_timeline.py analyse /tmp/lt_eager > $O/r06c_latency_timeline_eager.txt 2>&1
( cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/lt_graph -- python $GRAFT_REPO_ROOT/tools/latency_timeline.py run --steps 12 --graph ) > $O/r06c_latency_graph_prof.txt 2>&1
python tools/latency_timeline.py analyse /tmp/lt_graph > $O/r06c_latency_timeline_graph.txt 2>&1
cat $O/r06c_gemm_timeline.txt $O/r06c_pytest_graph.txt; tail -2 $O/r06c_latency_eager.txt $O/r06c_latency_graph.txt; cat $O/r06c_latency_timeline_eager.txt $O/r06c_latency_timeline_graph.txt
