#!/bin/bash
# Round-6 end-of-round validation on one MI355X: the full GPU suite, smoke(), the default bench command, the short bench under
# rocprofv3 (kernel trace + stats), the solve-stage timelines.  Small text files only -> gpurun_out/final6/
R=$PWD; O=$R/gpurun_out/final6; mkdir -p $O; export TMPDIR=/tmp
cd $R
SECONDS=0; python -m pytest tests -q -rs --durations=8 -m gpu > $O/suite.log 2>&1; echo "pytest rc=$? wall ${SECONDS}s" > $O/summary.txt
grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" $O/suite.log | tail -16 >> $O/summary.txt
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log >> $O/summary.txt
SECONDS=0; python bench.py > $O/bench.json 2> $O/bench.err; echo "bench.py rc=$? wall: $SECONDS s" >> $O/summary.txt; grep -i "PARITY" $O/bench.err | tail -3 >> $O/summary.txt
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/p_bench -o b -- python $R/bench.py --steps 3 --warmup 1 --no-extras --no-cpu-baseline > $O/bench_profiled.json 2> $O/bench_profiled.err
f=$(find /tmp/p_bench -name "*results.db" | head -1); python $R/tools/rocpd_stats.py $f > $O/bench_kernel_stats.md 2>&1; rm -rf /tmp/p_bench
cd $R; tools/r6_solve_timeline.sh > $O/timeline.log 2>&1; cp gpurun_out/soltl/timeline_rcca.md gpurun_out/soltl/timeline_mcca.md $O/ 2>/dev/null
cat $O/summary.txt
python - <<'PY'
import json
d = json.loads(open("gpurun_out/final6/bench.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step", "step_ms")}, d["roofline"]["frac"], d["roofline"]["achieved"], d["phases_ms"])
ex = d.get("extra", {})
print("k1", {k: (v.get("k1_ms"), v.get("k1_rel_err"), v.get("gram_frac_of_peak"), v.get("gram_stages_ms")) for k, v in ex.get("k1_routes", {}).items() if isinstance(v, dict)})
print("dcca_loss", {k: ex.get("dcca_loss", {}).get(k) for k in ("ms", "ms_sync_each")}, "metric shape", ex.get("dcca_loss_metric_shape", {}).get("ms"))
print("train", ex.get("dcca_training_step"))
for k, v in ex.get("configs", {}).items():
    print(k, {kk: v.get(kk) for kk in ("fit_ms", "gram_frac_of_peak", "solve_ms", "solve_ms_runs")})
print("evd", {k: (v.get("ms"), v.get("torch_eigh_ms", v.get("torch_svd_ms"))) for k, v in ex.get("dense_evd", {}).items() if isinstance(v, dict) and "ms" in v})
print("transform", ex.get("transform"))
pd = json.loads(open("gpurun_out/final6/bench_profiled.json").read().strip().splitlines()[-1])
print("profiled", pd["ms_per_step"], pd["roofline"])
PY
