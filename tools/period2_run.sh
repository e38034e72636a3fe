#!/bin/bash
# Period-2 solve slowdown: loop shapes and read-once switches on a pre-loaded box -> gpurun_out/r3a_p2_*.log
mkdir -p gpurun_out
python tools/period2_probe.py --preheat-s 75 --fits 14 --modes fit,short,gap,solve,two > gpurun_out/r3a_p2_default.log 2>&1
CCZ_POTRF_LOOKAHEAD=0 python tools/period2_probe.py --fits 14 --modes fit,solve > gpurun_out/r3a_p2_nolookahead.log 2>&1
CCZ_GRAPHS=0 python tools/period2_probe.py --fits 14 --modes fit > gpurun_out/r3a_p2_nographs.log 2>&1
python bench.py --steps 10 --warmup 2 --no-extras --no-cpu-baseline > gpurun_out/r3a_bench.json 2> gpurun_out/r3a_bench.err
grep -h "^##" gpurun_out/r3a_p2_*.log
python -c "import json;d=json.load(open('gpurun_out/r3a_bench.json'));print(d['step_ms'],d['phases_ms'])"
