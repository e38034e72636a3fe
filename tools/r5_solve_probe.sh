#!/bin/bash
# Round 5, solve stage: parity of the solver tests on the chain kernel, then A/B timings of the three solve shapes.
R=$PWD; O=$R/gpurun_out/${1:-r5s}; mkdir -p $O; export TMPDIR=/tmp
cd $R
timeout 1200 python -m pytest tests/test_gpu_round5.py tests/test_gpu_round2.py tests/test_gpu_estimators.py tests/test_gpu_ops.py -x -q -m gpu > $O/t_solver.log 2>&1; echo "tests rc=$?" > $O/summary.txt; tail -4 $O/t_solver.log >> $O/summary.txt
run() { name=$1; shift; echo "== $name $*" >> $O/summary.txt; env "$@" timeout 300 python tools/solve_probe.py rcca 4096,4096 64 100000 2>&1 | grep solve | tail -2 >> $O/summary.txt; }
run default CCZ_NOP=1
run nochain CCZ_CHOLINV_CHAIN=0
run nochain_p4 CCZ_CHOLINV_CHAIN=0 CCZ_CHOLINV_MFMA=1
run la32 CCZ_CHAIN_WGS_LA=32
run la128 CCZ_CHAIN_WGS_LA=128
run p4 CCZ_CHOLINV_MFMA=1
echo "== mcca 4x2048 default / nochain" >> $O/summary.txt
timeout 300 python tools/solve_probe.py mcca 2048,2048,2048,2048 64 100000 2>&1 | grep solve | tail -2 >> $O/summary.txt
CCZ_CHOLINV_CHAIN=0 timeout 300 python tools/solve_probe.py mcca 2048,2048,2048,2048 64 100000 2>&1 | grep solve | tail -2 >> $O/summary.txt
echo "== gcca 4096,4096,8192 k=128 default / nochain" >> $O/summary.txt
timeout 400 python tools/solve_probe.py gcca 4096,4096,8192 128 60000 2>&1 | grep solve | tail -2 >> $O/summary.txt
CCZ_CHOLINV_CHAIN=0 timeout 400 python tools/solve_probe.py gcca 4096,4096,8192 128 60000 2>&1 | grep solve | tail -2 >> $O/summary.txt
echo "== loss" >> $O/summary.txt
timeout 120 python tools/loss_profile.py 8192 512 40 2>/dev/null | tail -1 >> $O/summary.txt
CCZ_CHAIN_DEBUG=1 timeout 120 python tools/loss_profile.py 8192 512 2 2> $O/chain_debug.txt > /dev/null; tail -10 $O/chain_debug.txt >> $O/summary.txt
cat $O/summary.txt
