#!/bin/bash
# End-of-round validation on one MI355X: the full GPU suite, smoke(), the default bench command, the configs[3] loss profile.
# Everything judged is copied to gpurun_out/ (small text files only).
R=$PWD; mkdir -p gpurun_out; export TMPDIR=/tmp
python -m pytest tests -x -q -rs --durations=25 -m gpu > gpurun_out/final_pytest.log 2>&1; grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" gpurun_out/final_pytest.log | tail -40
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/final_smoke.log 2>&1; tail -1 gpurun_out/final_smoke.log
SECONDS=0; python bench.py > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err; tail -c 800 gpurun_out/final_bench.json; echo "bench.py wall: $SECONDS s"; grep -i "PARITY" gpurun_out/final_bench.err
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/p_loss -o loss -- python $R/tools/loss_profile.py 8192 512 20 > $R/gpurun_out/r03_loss_profile.log 2>&1
f=$(find /tmp/p_loss -name "*results.db" | head -1); python $R/tools/rocpd_stats.py $f > $R/gpurun_out/r03_loss_c4.md; tail -2 $R/gpurun_out/r03_loss_profile.log >> $R/gpurun_out/r03_loss_c4.md; rm -rf /tmp/p_loss
