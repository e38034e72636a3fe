#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_round3.py tests/test_gpu_ops.py -x -q -m gpu -k "blocks_layout or sharded_fits or transform or separated_spectrum" > gpurun_out/r3f_tests.log 2>&1; grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" gpurun_out/r3f_tests.log | tail -12
python tools/transform_probe.py > gpurun_out/r3f_transform.log 2>&1; tail -2 gpurun_out/r3f_transform.log
python tools/transform_probe.py 1000000 4096 16 > gpurun_out/r3f_transform16.log 2>&1; tail -2 gpurun_out/r3f_transform16.log
CCZ_GEMM_TALL_IMPL=0 python tools/transform_probe.py > gpurun_out/r3f_transform_old.log 2>&1; tail -2 gpurun_out/r3f_transform_old.log
CCZ_BENCH_FORCE_SHARDED=1 python bench.py --no-extras --no-cpu-baseline --steps 6 --warmup 2 > gpurun_out/r3f_bench_sharded.json 2> gpurun_out/r3f_bench_sharded.err
python -c "import json;d=json.load(open('gpurun_out/r3f_bench_sharded.json'));print('forced sharded', d['value'], d['step_ms'],d['phases_ms'], d['parity_gate']['ok'])" || tail -5 gpurun_out/r3f_bench_sharded.err
