#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_round3.py tests/test_gpu_round2.py tests/test_gpu_moments.py -x -q -m gpu -k "pilot or moments or offset or drifting or streamed" > gpurun_out/r3i_tests.log 2>&1; grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" gpurun_out/r3i_tests.log | tail -8
python tools/pilot_probe.py > gpurun_out/r3i_pilot.log 2>&1; cat gpurun_out/r3i_pilot.log | grep offset
for ov in 1 0; do
CCZ_COLSUM_OVERLAP=$ov python bench.py --no-extras --no-cpu-baseline --steps 8 --warmup 2 > gpurun_out/r3i_bench_ov$ov.json 2> gpurun_out/r3i_bench_ov$ov.err
python -c "import json;d=json.load(open('gpurun_out/r3i_bench_ov$ov.json'));print('overlap $ov', d['value'], d['step_ms'],d['phases_ms'], d['roofline']['frac'])" || tail -3 gpurun_out/r3i_bench_ov$ov.err
done
