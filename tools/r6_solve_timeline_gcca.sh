#!/bin/bash
# Round 6: kernel trace of one GCCA solve (D = 16384, k = 128) -> tools/solve_timeline.py -> gpurun_out/soltl/timeline_gcca.md
R=$PWD; O=$R/gpurun_out/soltl; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
rm -rf /tmp/p_sol_g
timeout 400 rocprofv3 --kernel-trace -d /tmp/p_sol_g -o s -- python $R/tools/solve_probe.py gcca 4096,4096,8192 128 60000 > $O/gcca.log 2>&1
f=$(find /tmp/p_sol_g -name "*results.db" | head -1)
[ -n "$f" ] && python $R/tools/solve_timeline.py "$f" > $O/timeline_gcca.md
tail -1 $O/gcca.log
