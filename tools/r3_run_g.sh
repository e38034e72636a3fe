#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_round3.py tests/test_gpu_round2.py tests/test_gpu_estimators.py tests/test_gpu_edge_cases.py -x -q -m gpu -k "sharded_fits or separated_spectrum or ns_shape_against_oracle or estimators or edge or pilot or offset_golden" > gpurun_out/r3g_tests.log 2>&1; grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" gpurun_out/r3g_tests.log | tail -12
for r in 1 0; do
CCZ_POTRF_RIDER=$r CCZ_TRACE_PHASES=2 python bench.py --no-extras --no-cpu-baseline --steps 8 --warmup 2 > gpurun_out/r3g_bench_rider$r.json 2> gpurun_out/r3g_bench_rider$r.err
python -c "import json;d=json.load(open('gpurun_out/r3g_bench_rider$r.json'));print('rider $r', d['value'], d['step_ms'],d['phases_ms'])"
grep "rcca phases" gpurun_out/r3g_bench_rider$r.err | tail -2
done
python tools/solve_probe.py rcca 4096,4096 64 200000 2>&1 | tail -4
python tools/solve_probe.py mcca 2048,2048,2048,2048 64 200000 2>&1 | tail -3
