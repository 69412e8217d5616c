#!/bin/bash
# round 5: the training-step tests not in tools/r05_dwconv_sinks.sh's selection, with the final defaults
O=gpurun_out; mkdir -p $O
( timeout 95 python -m pytest tests/test_gpu_train_step.py -x -q -m gpu -k "not (depthwise or weight_gradient_lanes or tape_equals or two_lanes_equal or replicas or matches_reference or host_sync)" 2>&1 | tail -6 ) > $O/r05an_pytest_rest.txt
tail -3 $O/r05an_pytest_rest.txt
