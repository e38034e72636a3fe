#!/bin/bash
# rocprofv3 kernel statistics of the blocked EVD (evd_block.hip): gpurun_out/evd_<d>_stats.md
R=$PWD; mkdir -p gpurun_out; export TMPDIR=/tmp; cd /tmp
for d in "$@"; do
  timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/p_evd_$d -o s -- python $R/tools/evd_probe.py syev $d 2 > $R/gpurun_out/evd_${d}.log 2>&1
  f=$(find /tmp/p_evd_$d -name "*results.db" | head -1)
  python $R/tools/rocpd_stats.py $f k_bj > $R/gpurun_out/evd_${d}_stats.md 2>&1
  rm -rf /tmp/p_evd_$d
  tail -2 $R/gpurun_out/evd_${d}.log
done
