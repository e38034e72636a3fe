mkdir -p gpurun_out; export TMPDIR=/tmp; R=$PWD
for v in plain sync gc; do VARIANT=$v timeout 120 python tools/bench_alt.py > gpurun_out/r2i_alt_$v.log 2>&1; done
cd /tmp
# kernel stats of the bench (short: 3 steps, no cpu baseline), loss C4, solve
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/p_bench -o bench -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $R/gpurun_out/r2i_bench_profiled.json 2> $R/gpurun_out/r2i_bench_profiled.err
f=$(find /tmp/p_bench -name "*results.db" | head -1); python $R/tools/rocpd_stats.py $f k_gram_f32_fifo k_gram_f64_fifo k_gemm_f32_nn_fifo k_cholinv_step > $R/gpurun_out/r2i_bench_kernel_stats.md; rm -rf /tmp/p_bench
timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/p_loss -o loss -- python $R/tools/loss_profile.py 8192 512 20 > $R/gpurun_out/r2i_loss_profile.log 2>&1
f=$(find /tmp/p_loss -name "*results.db" | head -1); python $R/tools/rocpd_stats.py $f > $R/gpurun_out/r2i_loss_kernel_stats.md; rm -rf /tmp/p_loss
timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/p_solve -o solve -- python $R/tools/solve_probe.py rcca 4096,4096 64 > $R/gpurun_out/r2i_solve_probe.log 2>&1
f=$(find /tmp/p_solve -name "*results.db" | head -1); python $R/tools/rocpd_stats.py $f > $R/gpurun_out/r2i_solve_kernel_stats.md; rm -rf /tmp/p_solve
# PMC passes of K1, one counter group per run
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $grp --kernel-trace -d /tmp/pmc32_$i -o p -- python $R/tools/gram_probe.py --n 262144 --d 4096 --views 2 --dtype f32 --iters 2 > $R/gpurun_out/r2i_pmc32_$i.log 2>&1
  timeout 200 rocprofv3 --pmc $grp --kernel-trace -d /tmp/pmc64_$i -o p -- python $R/tools/gram_probe.py --n 131072 --d 4096 --views 2 --dtype f64 --iters 2 > $R/gpurun_out/r2i_pmc64_$i.log 2>&1
done
python $R/tools/pmc_extract.py k_gram_f32_fifo $(find /tmp/pmc32_* -name "*results.db") > $R/gpurun_out/r2i_pmc32.md 2>&1
python $R/tools/pmc_extract.py k_gram_f64_fifo $(find /tmp/pmc64_* -name "*results.db") > $R/gpurun_out/r2i_pmc64.md 2>&1
rm -rf /tmp/pmc32_* /tmp/pmc64_*
cd $R
tail -3 gpurun_out/r2i_alt_plain.log; tail -2 gpurun_out/r2i_alt_sync.log; tail -2 gpurun_out/r2i_alt_gc.log; cat gpurun_out/r2i_pmc32.md gpurun_out/r2i_pmc64.md; grep iter gpurun_out/r2i_pmc32_1.log gpurun_out/r2i_pmc64_1.log
