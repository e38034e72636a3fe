#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_round3.py tests/test_gpu_moments.py tests/test_gpu_estimators.py -x -q -m gpu -k "not wide_cca and not metric_shape" > gpurun_out/r3c_tests.log 2>&1
tail -5 gpurun_out/r3c_tests.log
CCZ_TRACE_PHASES=2 python tools/period2_probe2.py plain 10 60 > gpurun_out/r3c_p2_trace.log 2>&1
grep -h "^##" gpurun_out/r3c_p2_trace.log
grep "rcca phases" gpurun_out/r3c_p2_trace.log | tail -6
python tools/loss_probe.py 8192 512 30 > gpurun_out/r3c_loss_probe.log 2>&1; tail -8 gpurun_out/r3c_loss_probe.log
python bench.py --no-cpu-baseline > gpurun_out/r3c_bench.json 2> gpurun_out/r3c_bench.err; tail -c 600 gpurun_out/r3c_bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r3c_bench.json'))
print(d['value'], d['step_ms'], d['phases_ms'])
e=d.get('extra',{})
for k,v in e.items():
    if k=='configs':
        for kk,vv in v.items(): print(kk, {x:vv[x] for x in vv if x not in ('config',)})
    else: print(k, v)
PY
