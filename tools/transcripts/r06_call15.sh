#!/bin/bash
# round 6, call 15: does the FFN pair run faster per row when the hidden activation of a row chunk fits the 256 MB Infinity Cache?
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out
for M in 8192 16384 24576 32768 49152 82688; do echo "== M $M"; python tools/gemm_bench.py --tile 2 --iters 60 --M $M --only ffn 2>&1 | grep -v amdgpu.ids; done > $O/r06m_ffn_rows_sweep.txt 2>&1
cat $O/r06m_ffn_rows_sweep.txt
