#!/bin/bash
# Round 6: the solve stage with the pipelined fp64 GEMM (k_gemm_f64_pipe) against the round-2 kernels (CCZ_GEMM64_PIPE=0),
# fresh process each (the switch is read once).  -> stdout
for pipe in 1 0; do
  echo "== CCZ_GEMM64_PIPE=$pipe"
  CCZ_GEMM64_PIPE=$pipe python tools/solve_probe.py rcca 4096,4096 64 100000 2>&1 | tail -2
  CCZ_GEMM64_PIPE=$pipe python tools/solve_probe.py rcca 1024,1024 32 100000 2>&1 | tail -1
  CCZ_GEMM64_PIPE=$pipe python tools/solve_probe.py mcca 2048,2048,2048,2048 64 100000 2>&1 | tail -2
  CCZ_GEMM64_PIPE=$pipe python tools/solve_probe.py gcca 4096,4096,8192 128 60000 2>&1 | tail -2
done
