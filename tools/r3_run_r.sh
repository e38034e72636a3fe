#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_estimators.py tests/test_gpu_ops.py tests/test_gpu_loss.py -x -q -m gpu > gpurun_out/r3r_tests.log 2>&1; grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" gpurun_out/r3r_tests.log | tail -3
CCZ_TRACE_PHASES=2 python bench.py --no-cpu-baseline > gpurun_out/r3r_bench.json 2> gpurun_out/r3r_bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r3r_bench.json'))
print(d['value'], d['step_ms'], d['phases_ms'])
for k,v in d['extra']['configs'].items(): print(k, round(v['fit_ms'],1), v['solve_ms_runs'], v['parity_gate']['ok'])
PY
grep "rcca phases" gpurun_out/r3r_bench.err | tail -8 | cut -c1-300 || true
