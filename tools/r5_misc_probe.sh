#!/bin/bash
R=$PWD; O=$R/gpurun_out/${1:-r5m}; mkdir -p $O; export TMPDIR=/tmp
cd $R
timeout 1500 python -m pytest tests/test_gpu_multi.py tests/test_gpu_seams_r5.py tests/test_gpu_round4.py tests/test_gpu_round5.py tests/test_gpu_loss.py tests/test_gpu_deep_next.py -q -m gpu --durations=12 > $O/t_misc.log 2>&1; echo "tests rc=$?" > $O/summary.txt; tail -24 $O/t_misc.log >> $O/summary.txt
for tr in torch ccz; do
  CCZ_BENCH_FORCE_SHARDED=1 timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --transport $tr > $O/r05_bench_forced_sharded_$tr.json 2> $O/bench_$tr.err; echo "bench $tr rc=$?" >> $O/summary.txt
  python - <<PY >> $O/summary.txt
import json
try:
    d = json.loads(open("$O/r05_bench_forced_sharded_$tr.json").read().strip().splitlines()[-1])
    print({k: d.get(k) for k in ("value", "ms_per_step", "transport", "rccl_ranks", "ccz_comm_ranks")}, d["phases_ms"])
except Exception as e:
    print("no json:", e)
PY
done
cat $O/summary.txt
