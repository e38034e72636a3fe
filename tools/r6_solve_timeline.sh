#!/bin/bash
# Round 6: kernel trace of the rCCA / MCCA solve stage -> tools/solve_timeline.py -> profiles/r06_solve_timeline_*.md
R=$PWD; O=$R/gpurun_out/soltl; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
for kind in rcca mcca; do
  dims=4096,4096; [ $kind = mcca ] && dims=2048,2048,2048,2048
  rm -rf /tmp/p_sol_$kind
  timeout 300 rocprofv3 --kernel-trace -d /tmp/p_sol_$kind -o s -- python $R/tools/solve_probe.py $kind $dims 64 100000 > $O/$kind.log 2>&1
  f=$(find /tmp/p_sol_$kind -name "*results.db" | head -1)
  [ -n "$f" ] && python $R/tools/solve_timeline.py "$f" > $O/timeline_$kind.md
  tail -2 $O/$kind.log
done
