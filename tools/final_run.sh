#!/bin/bash
# End-of-round validation on one MI355X: the full GPU suite, the default bench, the bench under rocprofv3 (kernel
# statistics) and the rCCA solve profile.  Everything judged is copied to gpurun_out/ (small text files only).
R=$PWD; mkdir -p gpurun_out; export TMPDIR=/tmp
python -m pytest tests -x -q -m gpu > gpurun_out/final_pytest.log 2>&1; tail -3 gpurun_out/final_pytest.log
python bench.py > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err; tail -c 1500 gpurun_out/final_bench.json; tail -3 gpurun_out/final_bench.err
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/p_bench -o bench -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $R/gpurun_out/final_bench_profiled.json 2> $R/gpurun_out/final_bench_profiled.err
f=$(find /tmp/p_bench -name "*results.db" | head -1); python $R/tools/rocpd_stats.py $f k_gram_f32_fifo k_gram_f64_fifo k_gemm_f32_nn_fifo k_cholinv_step > $R/gpurun_out/final_bench_kernel_stats.md; rm -rf /tmp/p_bench
timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/p_loss -o loss -- python $R/tools/loss_profile.py 8192 512 20 > $R/gpurun_out/final_loss_profile.log 2>&1
f=$(find /tmp/p_loss -name "*results.db" | head -1); python $R/tools/rocpd_stats.py $f > $R/gpurun_out/final_loss_kernel_stats.md; rm -rf /tmp/p_loss
cd $R; tools/solve_prof.sh
