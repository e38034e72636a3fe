#!/bin/bash
# Round-6 evidence (one MI355X): PMC passes of the split-bf16 K1 (one counter group per run, never combined with trace domains),
# DVFS check, kernel statistics of a short bench run, of the metric-shape loss and of the configs[3] loss.  -> gpurun_out/r6p/
R=$PWD; O=$R/gpurun_out/r6p; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
N=262144
(cd $R && bash tools/r6_k1_profiles.sh $N) > $O/k1_profiles.log 2>&1
python $R/tools/pmc_to_traffic.py $O/r06_k_gram_bf16x2_pmc_raw.md $N 8192 > $O/r06_gram_traffic.json 2>> $O/k1_profiles.log
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/p_bench -o b -- python $R/bench.py --steps 3 --warmup 1 --no-extras --no-cpu-baseline > $O/bench_profiled.json 2> $O/bench_profiled.err
f=$(find /tmp/p_bench -name "*results.db" | head -1); python $R/tools/rocpd_stats.py $f > $O/r06_bench_kernel_stats.md 2>&1; rm -rf /tmp/p_bench
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/p_lossm -o l -- python $R/tools/loss_profile.py 1000000 4096 3 > $O/loss_metric.log 2>&1
f=$(find /tmp/p_lossm -name "*results.db" | head -1); python $R/tools/rocpd_stats.py $f > $O/r06_loss_metric_shape.md 2>&1; tail -2 $O/loss_metric.log >> $O/r06_loss_metric_shape.md; rm -rf /tmp/p_lossm
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/p_loss -o loss -- python $R/tools/loss_profile.py 8192 512 20 > $O/loss_profile.log 2>&1
f=$(find /tmp/p_loss -name "*results.db" | head -1); python $R/tools/rocpd_stats.py $f > $O/r06_loss_c4.md; tail -2 $O/loss_profile.log >> $O/r06_loss_c4.md; rm -rf /tmp/p_loss
cd $R; head -30 $O/r06_bench_kernel_stats.md; head -24 $O/r06_loss_metric_shape.md; cat $O/r06_gram_traffic.json
