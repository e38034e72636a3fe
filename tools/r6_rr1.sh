#!/bin/bash
# Round 6: the first Rayleigh-Ritz of the subspace iteration with a Jacobi threshold of 1e-9 (default) against a full-accuracy
# one (CCZ_RR1_TOL=0), fresh process each (the switch is read once); CCZ_TRACE_SOLVER prints the sweeps and the cycles.  -> stdout
for tol in 1e-9 0 1e-6; do
  echo "== CCZ_RR1_TOL=$tol"
  CCZ_RR1_TOL=$tol CCZ_TRACE_SOLVER=1 python tools/solve_probe.py rcca 4096,4096 64 100000 2>&1 | grep -v amdgpu.ids | tail -7
  CCZ_RR1_TOL=$tol CCZ_TRACE_SOLVER=1 python tools/solve_probe.py rcca 1024,1024 32 100000 2>&1 | grep -v amdgpu.ids | tail -5
  CCZ_RR1_TOL=$tol CCZ_TRACE_SOLVER=1 python tools/solve_probe.py mcca 2048,2048,2048,2048 64 100000 2>&1 | grep -v amdgpu.ids | tail -5
  CCZ_RR1_TOL=$tol CCZ_TRACE_SOLVER=1 python tools/solve_probe.py gcca 4096,4096,8192 128 60000 2>&1 | grep -v amdgpu.ids | tail -5
done
