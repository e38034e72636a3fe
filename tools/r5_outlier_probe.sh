#!/bin/bash
# Where do the 16 / 25 ms of an outlier solve inside bench.py's extras go?  Phases per solve (CCZ_TRACE_PHASES=1), pool misses.
R=$PWD; O=$R/gpurun_out/${1:-r5o}; mkdir -p $O; export TMPDIR=/tmp; cd $R
CCZ_TRACE_PHASES=1 timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --only configs --no-gates > $O/bench_chain.json 2> $O/bench_chain.err
grep "phases" $O/bench_chain.err | tail -9; python - <<PY
import json
d = json.loads(open("$O/bench_chain.json").read().strip().splitlines()[-1])
for k, v in d.get("extra", {}).get("configs", {}).items():
    print(k, v.get("solve_ms_runs"))
PY
