#!/bin/bash
for env in "CCZ_NOP=1" "CCZ_AUX_PRIORITY=0" "CCZ_GEMM_BIG_MIN_TILES=129" "CCZ_GEMM_BIG_MIN_TILES=257"; do
echo "== $env"
env $env python tools/solve_probe.py rcca 4096,4096 64 200000 2>&1 | tail -2
env $env python tools/solve_probe.py mcca 2048,2048,2048,2048 64 200000 2>&1 | tail -1
done
echo "== gcca default / no priority"
python tools/solve_probe.py gcca 4096,4096,8192 128 60000 2>&1 | tail -1
CCZ_AUX_PRIORITY=0 python tools/solve_probe.py gcca 4096,4096,8192 128 60000 2>&1 | tail -1
