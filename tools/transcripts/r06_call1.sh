#!/bin/bash
# round 6, call 1: the world-size-8 tests + touched tests, then the default bench line (today's baseline with the box calibration)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
python -m pytest tests/test_gpu_multiprocess.py -x -q -m gpu -s 2>&1 | tail -40 > gpurun_out/r06a_pytest_multiprocess.txt
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -s -k "fullsize_batch" 2>&1 | grep -v "^$" | tail -30 > gpurun_out/r06a_pytest_fullsize.txt
python -m pytest tests/test_gpu_train_step.py -x -q -m gpu -k "cross_entropy or weight_gradient_lanes or two_rank" 2>&1 | tail -8 > gpurun_out/r06a_pytest_train.txt
python bench.py > gpurun_out/r06a_bench.json 2> gpurun_out/r06a_bench.err
tail -c 600 gpurun_out/r06a_bench.err
tail -5 gpurun_out/r06a_pytest_multiprocess.txt gpurun_out/r06a_pytest_fullsize.txt gpurun_out/r06a_pytest_train.txt
