#!/bin/bash
# Round 5: kernel + memory-copy trace of CCA.fit on pinned HOST views (tools/host_fit_probe2.py) -- the timeline behind
# bench.py's extra.host_inputs -> tools/host_timeline.py -> profiles/r05_host_fit_timeline.md;
# plus a sanity pass of the moments / pipeline tests on the build that is shipped.
R=$PWD; O=$R/gpurun_out/hostfit; mkdir -p $O; export TMPDIR=/tmp; cd $R
python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -2 > $O/summary.txt
timeout 600 python -m pytest tests/test_gpu_moments.py tests/test_gpu_round5.py tests/test_gpu_estimators.py -q -x -m gpu 2>&1 | grep -E "passed|failed|rror" | tail -3 >> $O/summary.txt
cd /tmp
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace -d /tmp/p_host -o h -- python $R/tools/host_fit_probe2.py > $O/probe.log 2>&1
f=$(find /tmp/p_host -name "*results.db" | head -1); [ -n "$f" ] && cp "$f" $O/h_results.db
grep fit_s $O/probe.log >> $O/summary.txt
cd $R; cat $O/summary.txt
