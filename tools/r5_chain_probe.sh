#!/bin/bash
# Round 5: chain-kernel variants at configs[3] (grid size, poll sleep, panel form) + shader-clock stamps of both forms.
R=$PWD; O=$R/gpurun_out/${1:-r5d}; mkdir -p $O; export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests/test_gpu_round5.py -x -q -m gpu > $O/t_round5.log 2>&1; echo "round5 rc=$?" > $O/summary.txt; tail -3 $O/t_round5.log >> $O/summary.txt
run() { name=$1; shift; echo -n "$name: " >> $O/summary.txt; env "$@" timeout 120 python tools/loss_profile.py 8192 512 40 2> $O/lp_$name.err | tail -1 >> $O/summary.txt; }
run default CCZ_DUMMY=1
run default_again CCZ_DUMMY=1
run p4 CCZ_CHOLINV_MFMA=1
run nochain CCZ_CHOLINV_CHAIN=0
run nofast CCZ_LOSS_FAST=0
run k1staged CCZ_LOSS_K1_FIFO=0
for w in 64 200; do run wgs$w CCZ_CHAIN_WGS=$w; done
for sl in 4; do run sleep$sl CCZ_CHAIN_SLEEP=$sl; done
run wgs64_sleep4 CCZ_CHAIN_WGS=64 CCZ_CHAIN_SLEEP=4
CCZ_CHAIN_DEBUG=1 timeout 120 python tools/loss_profile.py 8192 512 2 2> $O/chain_debug_A.txt > /dev/null; tail -10 $O/chain_debug_p16.txt >> $O/summary.txt
CCZ_CHOLINV_MFMA=1 CCZ_CHAIN_DEBUG=1 timeout 120 python tools/loss_profile.py 8192 512 2 2> $O/chain_debug_B.txt > /dev/null; tail -10 $O/chain_debug_p4.txt >> $O/summary.txt
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/p_loss -o loss -- python $R/tools/loss_profile.py 8192 512 20 > $O/loss_profile.log 2>&1
f=$(find /tmp/p_loss -name "*results.db" | head -1); python $R/tools/rocpd_stats.py $f > $O/r05_loss_c4.md; tail -2 $O/loss_profile.log >> $O/r05_loss_c4.md; rm -rf /tmp/p_loss
cd $R; head -16 $O/r05_loss_c4.md >> $O/summary.txt
cat $O/summary.txt
