#!/bin/bash
# End-of-round validation on one MI355X: the full GPU suite, smoke(), the default bench command, the bench command under
# rocprofv3 (kernel trace).  Everything judged is copied to gpurun_out/final/ (small text files only).
R=$PWD; O=$R/gpurun_out/final; mkdir -p $O; export TMPDIR=/tmp
cd $R
SECONDS=0; python -m pytest tests -q -rs --durations=12 -m gpu > $O/final_pytest.log 2>&1; echo "pytest rc=$? wall ${SECONDS}s" > $O/summary.txt
grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" $O/final_pytest.log | tail -30 >> $O/summary.txt
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $O/final_smoke.log 2>&1; tail -2 $O/final_smoke.log >> $O/summary.txt
SECONDS=0; python bench.py > $O/final_bench.json 2> $O/final_bench.err; echo "bench.py rc=$? wall: $SECONDS s" >> $O/summary.txt; grep -i "PARITY" $O/final_bench.err >> $O/summary.txt
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/p_bench -o bench -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extras > $O/bench_profiled.json 2> $O/bench_profiled.err
f=$(find /tmp/p_bench -name "*results.db" | head -1); python $R/tools/rocpd_stats.py $f k_gram_f32_fifo k_colsum > $O/bench_kernel_stats.md; rm -rf /tmp/p_bench
cd $R; cat $O/summary.txt; python - <<'PY'
import json
d = json.loads(open("gpurun_out/final/final_bench.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step", "step_ms")}, d["roofline"]["frac"], d["phases_ms"])
ex = d.get("extra", {})
print("dcca_loss", {k: ex.get("dcca_loss", {}).get(k) for k in ("ms", "ms_sync_each", "torch_gpu_ms")})
print("train", ex.get("dcca_training_step"))
for k, v in ex.get("configs", {}).items():
    print(k, {kk: v.get(kk) for kk in ("fit_ms", "gram_frac_of_peak", "solve_ms", "solve_ms_runs")})
print("evd", {k: (v.get("ms"), v.get("torch_eigh_ms", v.get("torch_svd_ms"))) for k, v in ex.get("dense_evd", {}).items() if isinstance(v, dict) and "ms" in v})
c2 = d.get("cpu_baseline", {}).get("c2_rcca", {})
print("c2", c2.get("cpu_s"), c2.get("weights_vs_oracle"), c2.get("separated", {}).get("weights_vs_oracle"))
PY
