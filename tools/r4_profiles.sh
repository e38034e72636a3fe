#!/bin/bash
# Round-4 evidence: rocprofv3 kernel statistics of the bench command and of the configs[3] loss (the EVD statistics and
# PMC passes come from tools/evd_prof.sh / tools/evd_pmc.sh).  Small text files -> gpurun_out/r04_*, copied to profiles/.
R=$PWD; mkdir -p gpurun_out; export TMPDIR=/tmp; cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/p_bench -o bench -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extras > $R/gpurun_out/r04_bench_profiled.json 2> $R/gpurun_out/r04_bench_profiled.err
f=$(find /tmp/p_bench -name "*results.db" | head -1); python $R/tools/rocpd_stats.py $f k_gram_f32_fifo k_colsum > $R/gpurun_out/r04_bench_kernel_stats.md; rm -rf /tmp/p_bench
timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/p_loss -o loss -- python $R/tools/loss_profile.py 8192 512 20 > $R/gpurun_out/r04_loss_profile.log 2>&1
f=$(find /tmp/p_loss -name "*results.db" | head -1); python $R/tools/rocpd_stats.py $f > $R/gpurun_out/r04_loss_c4.md; tail -2 $R/gpurun_out/r04_loss_profile.log >> $R/gpurun_out/r04_loss_c4.md; rm -rf /tmp/p_loss
cd $R; head -12 gpurun_out/r04_bench_kernel_stats.md; head -8 gpurun_out/r04_loss_c4.md
