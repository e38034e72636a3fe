#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_estimators.py tests/test_gpu_ops.py tests/test_gpu_edge_cases.py tests/test_gpu_partial_group.py tests/test_gpu_model_selection.py tests/test_gpu_deep_next.py -x -q -m gpu > gpurun_out/r3n_tests.log 2>&1; grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" gpurun_out/r3n_tests.log | tail -4
python tools/solve_probe.py rcca 4096,4096 64 200000 2>&1 | tail -3
python tools/solve_probe.py rcca 1024,1024 32 100000 2>&1 | tail -3
python bench.py --no-extras --no-cpu-baseline --steps 8 --warmup 2 > gpurun_out/r3n_bench.json 2> gpurun_out/r3n_bench.err
python -c "import json;d=json.load(open('gpurun_out/r3n_bench.json'));print(d['value'], d['step_ms'],d['phases_ms'])"
tools/r3_run_m.sh
