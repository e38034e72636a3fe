#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_round3.py tests/test_gpu_loss.py tests/test_gpu_deep_next.py tests/test_gpu_round2.py -x -q -m gpu > gpurun_out/r3b_tests.log 2>&1
tail -15 gpurun_out/r3b_tests.log
tools/period2_run2.sh
