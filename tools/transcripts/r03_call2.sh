#!/bin/bash
# round 3, GPU call 2: whole GPU suite (new determinism gate, sinks under data parallelism) + default bench
O=gpurun_out/r03b; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q -s -k "fullsize_batch_vs_reference_golden" > $O/fullsize_parity_log.txt 2>&1
timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1
tail -15 $O/pytest_gpu.txt
grep "32 x 30 s" $O/fullsize_parity_log.txt
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 1500 $O/bench.json
