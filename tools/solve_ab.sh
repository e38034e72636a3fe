#!/bin/bash
# A/B timing of the solve stage: look-ahead Cholesky and two-sided Rayleigh-Ritz Jacobi on / off (fresh process each,
# the switches are read once).  Usage: tools/solve_ab.sh [n]   -> gpurun_out/solve_ab.log
n=${1:-200000}
mkdir -p gpurun_out
log=gpurun_out/solve_ab.log
: > $log
run() {
  echo "== $*" >> $log
  env "$@" python tools/solve_probe.py rcca 4096,4096 64 $n 2>&1 | tail -3 >> $log
}
run CCZ_NOP=1
run CCZ_POTRF_LOOKAHEAD=0
run CCZ_SYEV_TWOSIDED=0
run CCZ_POTRF_LOOKAHEAD=0 CCZ_SYEV_TWOSIDED=0
echo "== mcca 4x2048 (default / no look-ahead)" >> $log
python tools/solve_probe.py mcca 2048,2048,2048,2048 64 $n 2>&1 | tail -2 >> $log
CCZ_POTRF_LOOKAHEAD=0 python tools/solve_probe.py mcca 2048,2048,2048,2048 64 $n 2>&1 | tail -2 >> $log
echo "== gcca 4096,4096,8192 k=128 (default / no look-ahead)" >> $log
python tools/solve_probe.py gcca 4096,4096,8192 128 60000 2>&1 | tail -2 >> $log
CCZ_POTRF_LOOKAHEAD=0 python tools/solve_probe.py gcca 4096,4096,8192 128 60000 2>&1 | tail -2 >> $log
cat $log
