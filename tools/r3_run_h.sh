#!/bin/bash
# solve kernel profile (rCCA) with per-grid breakdown of the fp64 GEMMs + projection kernels at small d
R=$PWD; mkdir -p gpurun_out; export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/p_rcca -o s -- python $R/tools/solve_probe.py rcca 4096,4096 64 200000 > $R/gpurun_out/r3h_solve_rcca.log 2>&1
f=$(find /tmp/p_rcca -name "*results.db" | head -1)
python $R/tools/rocpd_stats.py $f k_gemm_f64_big k_gemm_f64_half k_gemm_f64_multi k_gemm_f64_skinny > $R/gpurun_out/r3h_solve_rcca_stats.md 2>&1
rm -rf /tmp/p_rcca
cd $R
head -40 gpurun_out/r3h_solve_rcca_stats.md
grep -A40 "launch shapes" gpurun_out/r3h_solve_rcca_stats.md | head -70
for impl in 1 0; do
echo "tall impl $impl"
CCZ_GEMM_TALL_IMPL=$impl python tools/transform_probe.py 4000000 512 64 2>&1 | grep transform
CCZ_GEMM_TALL_IMPL=$impl python tools/transform_probe.py 2000000 1024 64 2>&1 | grep transform
done
