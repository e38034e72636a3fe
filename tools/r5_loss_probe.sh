#!/bin/bash
# Round 5, loss path: parity of the new kernels first, then the configs[3] timings of every A/B switch, then the kernel
# trace.  Everything lands in gpurun_out/r5c/ (small text files).
R=$PWD; O=$R/gpurun_out/r5c; mkdir -p $O; export TMPDIR=/tmp
cd $R
echo "== round-5 tests" > $O/summary.txt
timeout 1200 python -m pytest tests/test_gpu_round5.py -x -q -m gpu --durations=8 > $O/t_round5.log 2>&1; echo "round5 rc=$?" >> $O/summary.txt
tail -25 $O/t_round5.log >> $O/summary.txt
timeout 900 python -m pytest tests/test_gpu_round2.py tests/test_gpu_loss.py tests/test_gpu_deep_next.py -x -q -m gpu -k "chol or loss or Loss or potrf or whiten" > $O/t_loss_old.log 2>&1; echo "older loss/chol tests rc=$?" >> $O/summary.txt
tail -5 $O/t_loss_old.log >> $O/summary.txt
echo "== timings (python tools/loss_profile.py 8192 512 40: median ms per fwd+bwd)" >> $O/summary.txt
run() { name=$1; shift; echo -n "$name: " >> $O/summary.txt; env "$@" timeout 120 python tools/loss_profile.py 8192 512 40 2> $O/lp_$name.err | tail -1 >> $O/summary.txt; }
run default CCZ_DUMMY=1
run legacy_all CCZ_LOSS_FAST=0 CCZ_CHOLINV_CHAIN=0 CCZ_CHOLINV_MFMA=1
run nochain CCZ_CHOLINV_CHAIN=0
run nochain_p4 CCZ_CHOLINV_CHAIN=0 CCZ_CHOLINV_MFMA=1
run chain_p4 CCZ_CHOLINV_MFMA=1
run nofast CCZ_LOSS_FAST=0
run splitk1 CCZ_LOSS_SPLITK=1
run splitk4 CCZ_LOSS_SPLITK=4
run wgs64 CCZ_CHAIN_WGS=64
run wgs250 CCZ_CHAIN_WGS=250
CCZ_CHAIN_DEBUG=1 timeout 120 python tools/loss_profile.py 8192 512 3 2> $O/chain_debug.txt > /dev/null; tail -12 $O/chain_debug.txt >> $O/summary.txt
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/p_loss -o loss -- python $R/tools/loss_profile.py 8192 512 20 > $O/loss_profile.log 2>&1
f=$(find /tmp/p_loss -name "*results.db" | head -1); python $R/tools/rocpd_stats.py $f > $O/r05_loss_c4.md; tail -2 $O/loss_profile.log >> $O/r05_loss_c4.md; rm -rf /tmp/p_loss
cd $R; head -40 $O/r05_loss_c4.md >> $O/summary.txt
cat $O/summary.txt
