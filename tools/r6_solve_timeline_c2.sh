R=$PWD; O=$R/gpurun_out/soltl; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
rm -rf /tmp/p_sol_c2
timeout 300 rocprofv3 --kernel-trace -d /tmp/p_sol_c2 -o s -- python $R/tools/solve_probe.py rcca 1024,1024 32 100000 > $O/c2.log 2>&1
f=$(find /tmp/p_sol_c2 -name "*results.db" | head -1)
[ -n "$f" ] && python $R/tools/solve_timeline.py "$f" > $O/timeline_c2.md
tail -2 $O/c2.log
