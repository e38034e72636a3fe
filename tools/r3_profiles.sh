#!/bin/bash
# Round-3 evidence: rocprofv3 kernel statistics of the bench command, the three solves and the configs[3] loss, and the PMC
# passes of K1 (plain and pilot-shifted FIFO kernel).  Small text files only -> gpurun_out/r03_*, copied to profiles/.
R=$PWD; mkdir -p gpurun_out; export TMPDIR=/tmp; cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/p_bench -o bench -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extras > $R/gpurun_out/r03_bench_profiled.json 2> $R/gpurun_out/r03_bench_profiled.err
f=$(find /tmp/p_bench -name "*results.db" | head -1); python $R/tools/rocpd_stats.py $f k_gram_f32_fifo k_colsum > $R/gpurun_out/r03_bench_kernel_stats.md; rm -rf /tmp/p_bench
prof() {  # name, args...
  local name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/p_$name -o s -- python $R/tools/solve_probe.py "$@" > $R/gpurun_out/r03_solve_${name}.log 2>&1
  f=$(find /tmp/p_$name -name "*results.db" | head -1)
  python $R/tools/rocpd_stats.py $f k_gemm_f64_big k_gemm_f64_half k_gemm_f64_skinny k_jacobi k_syev > $R/gpurun_out/r03_solve_${name}.md 2>&1
  grep solve $R/gpurun_out/r03_solve_${name}.log | tail -3 >> $R/gpurun_out/r03_solve_${name}.md
  rm -rf /tmp/p_$name
}
prof rcca rcca 4096,4096 64 200000
prof mcca mcca 2048,2048,2048,2048 64 200000
prof gcca gcca 4096,4096,8192 128 60000
timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/p_loss -o loss -- python $R/tools/loss_profile.py 8192 512 20 > $R/gpurun_out/r03_loss_profile.log 2>&1
f=$(find /tmp/p_loss -name "*results.db" | head -1); python $R/tools/rocpd_stats.py $f > $R/gpurun_out/r03_loss_c4.md; tail -2 $R/gpurun_out/r03_loss_profile.log >> $R/gpurun_out/r03_loss_c4.md; rm -rf /tmp/p_loss
for variant in plain pilot; do
  off=0; [ $variant = pilot ] && off=10
  i=0
  for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum"; do
    i=$((i+1))
    timeout 200 rocprofv3 --pmc $grp --kernel-trace -d /tmp/pmc_${variant}_$i -o p -- python $R/tools/gram_probe.py --n 262144 --d 4096 --views 2 --dtype f32 --iters 2 --offset $off > $R/gpurun_out/r03_pmc_${variant}_$i.log 2>&1
  done
  python $R/tools/pmc_extract.py k_gram_f32_fifo $(find /tmp/pmc_${variant}_* -name "*results.db") > $R/gpurun_out/r03_gram_pmc_${variant}.md 2>&1
  grep iter $R/gpurun_out/r03_pmc_${variant}_1.log >> $R/gpurun_out/r03_gram_pmc_${variant}.md
  rm -rf /tmp/pmc_${variant}_*
done
cd $R
cat gpurun_out/r03_gram_pmc_plain.md gpurun_out/r03_gram_pmc_pilot.md
head -12 gpurun_out/r03_bench_kernel_stats.md
