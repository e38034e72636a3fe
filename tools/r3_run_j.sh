#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_round3.py tests/test_gpu_round2.py tests/test_gpu_ops.py -x -q -m gpu -k "gemm_f64 or potrf or trsm or cholesky or c3_mcca or c5_gcca or ns_shape_against_oracle" > gpurun_out/r3j_tests.log 2>&1; grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" gpurun_out/r3j_tests.log | tail -8
for nt in 1 0; do
echo "NT_FIFO=$nt"
CCZ_GEMM_NT_FIFO=$nt python tools/solve_probe.py rcca 4096,4096 64 200000 2>&1 | tail -2
CCZ_GEMM_NT_FIFO=$nt python tools/solve_probe.py mcca 2048,2048,2048,2048 64 200000 2>&1 | tail -1
CCZ_GEMM_NT_FIFO=$nt python tools/solve_probe.py gcca 4096,4096,8192 128 60000 2>&1 | tail -1
done
CCZ_TRACE_PHASES=2 python tools/solve_probe.py rcca 4096,4096 64 200000 2>&1 | grep phases | tail -1
