#!/bin/bash
# Round 6: the piped split-route launch (split pass of the next row piece on a CU-masked side stream under the MFMA kernel of the
# current one) against the one-piece launch at the metric shape, over piece fractions and side-stream widths.  -> stdout
run() {
  echo "== CCZ_SPLIT_PIPE=$1 CCZ_SPLIT_PIPE_CUS=$2 n=$3"
  CCZ_SPLIT_PIPE=$1 CCZ_SPLIT_PIPE_CUS=$2 timeout 300 python tools/k1_route_check.py big $3 4096 2>&1 | grep '^{' | grep bf16x2
}
python tools/k1_route_check.py small 2>&1 | tail -3
run 0.125 64 262144
N=${1:-1000000}
run 0 64 $N
run 0.125 64 $N
run 0.125 0 $N
run 0.125 32 $N
run 0.125 128 $N
run 0.03,0.2 64 $N
run 0.06 64 $N
run 0.25 64 $N
