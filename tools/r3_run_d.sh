#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_round3.py tests/test_gpu_loss.py tests/test_gpu_estimators.py tests/test_gpu_deep_next.py -x -q -m gpu -k "not wide_cca and not metric_shape" > gpurun_out/r3d_tests.log 2>&1
tail -5 gpurun_out/r3d_tests.log
timeout 600 python -m pytest tests/test_gpu_round2.py tests/test_gpu_moments.py -x -q -m gpu -k "pilot or moments or offset" > gpurun_out/r3d_tests2.log 2>&1
tail -3 gpurun_out/r3d_tests2.log
python tools/pilot_probe.py > gpurun_out/r3d_pilot.log 2>&1; cat gpurun_out/r3d_pilot.log
CCZ_GRAM_FIFO_PILOT=0 python tools/pilot_probe.py > gpurun_out/r3d_pilot_staged.log 2>&1; grep "offset  10" gpurun_out/r3d_pilot_staged.log
CCZ_TRACE_PHASES=2 python bench.py --no-extras --no-cpu-baseline --steps 10 --warmup 2 > gpurun_out/r3d_bench_trace.json 2> gpurun_out/r3d_bench_trace.err
python -c "import json;d=json.load(open('gpurun_out/r3d_bench_trace.json'));print(d['step_ms'],d['phases_ms'])"
grep "rcca phases" gpurun_out/r3d_bench_trace.err | tail -10
python - <<'PY' > gpurun_out/r3d_loss.log 2>&1
import sys; sys.path.insert(0,'.')
import bench, json
print(json.dumps(bench.dcca_extra(gate=False)))
print(json.dumps(bench.training_step_extra()))
PY
cat gpurun_out/r3d_loss.log
