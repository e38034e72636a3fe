#!/bin/bash
mkdir -p gpurun_out
SECONDS=0; python bench.py > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err; echo "bench.py wall: $SECONDS s"; tail -c 600 gpurun_out/final_bench.json; grep -i "PARITY\|Error\|Traceback" gpurun_out/final_bench.err | head
timeout 900 python -m pytest tests/test_gpu_round3.py tests/test_gpu_round2.py -x -q -rs -m gpu -k "separated_spectrum or metric_shape or ns_shape_against" > gpurun_out/r3l_tests.log 2>&1; grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" gpurun_out/r3l_tests.log | tail -5
