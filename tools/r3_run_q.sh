#!/bin/bash
mkdir -p gpurun_out
CCZ_TRACE_PHASES=2 CCZ_TRACE_POOL=1 python bench.py --no-cpu-baseline > gpurun_out/r3q_bench.json 2> gpurun_out/r3q_bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r3q_bench.json'))
for k,v in d['extra']['configs'].items(): print(k, round(v['fit_ms'],1), v['solve_ms_runs'])
PY
grep -n "rcca phases\|pool miss\|graph miss" gpurun_out/r3q_bench.err | tail -14 | cut -c1-420
grep -c "pool miss" gpurun_out/r3q_bench.err; grep -c "graph miss" gpurun_out/r3q_bench.err
