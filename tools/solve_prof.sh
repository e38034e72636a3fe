#!/bin/bash
# rocprofv3 kernel statistics of the three solve drivers (3 solves each, after K1 at a small n) -> gpurun_out/solve_*_stats.md
R=$PWD; mkdir -p gpurun_out; export TMPDIR=/tmp; cd /tmp
prof() {  # name, args...
  local name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/p_$name -o s -- python $R/tools/solve_probe.py "$@" > $R/gpurun_out/solve_${name}.log 2>&1
  f=$(find /tmp/p_$name -name "*results.db" | head -1)
  python $R/tools/rocpd_stats.py $f > $R/gpurun_out/solve_${name}_stats.md 2>&1
  rm -rf /tmp/p_$name
  grep solve $R/gpurun_out/solve_${name}.log | tail -3
}
prof rcca rcca 4096,4096 64 200000
prof mcca mcca 2048,2048,2048,2048 64 200000
prof gcca gcca 4096,4096,8192 128 60000
