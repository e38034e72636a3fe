#!/bin/bash
mkdir -p gpurun_out
for spin in 50 0; do
  CCZ_SPIN_WAIT_MS=$spin CCZ_TRACE_PHASES=2 python bench.py --no-extras --no-cpu-baseline --steps 10 --warmup 2 > gpurun_out/r3e_bench_spin$spin.json 2> gpurun_out/r3e_bench_spin$spin.err
  python -c "import json;d=json.load(open('gpurun_out/r3e_bench_spin$spin.json'));print('spin $spin', d['step_ms'],d['phases_ms'])"
  grep "rcca phases" gpurun_out/r3e_bench_spin$spin.err | tail -4 | sed 's/.*|| /   /'
done
python bench.py --no-extras --no-cpu-baseline --steps 10 --warmup 2 > gpurun_out/r3e_bench_plain.json 2> gpurun_out/r3e_bench_plain.err
python -c "import json;d=json.load(open('gpurun_out/r3e_bench_plain.json'));print('plain', d['value'], d['step_ms'],d['phases_ms'])"
timeout 900 python -m pytest tests/test_gpu_round3.py -x -q -m gpu -k "separated_spectrum" > gpurun_out/r3e_tests_ns.log 2>&1; tail -15 gpurun_out/r3e_tests_ns.log
timeout 1500 python -m pytest tests -x -q -m gpu --deselect tests/test_gpu_round3.py::test_ns_dimensions_per_column_on_a_separated_spectrum > gpurun_out/r3e_tests_all.log 2>&1; tail -5 gpurun_out/r3e_tests_all.log
