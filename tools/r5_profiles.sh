#!/bin/bash
# Round-5 evidence: a fresh PMC pass of K1 (one counter group per run, never combined with trace domains), the split-bf16
# feasibility probe, kernel statistics of the configs[3] loss and of the three solve drivers.  -> gpurun_out/r5p/
R=$PWD; O=$R/gpurun_out/r5p; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $grp --kernel-trace -d /tmp/pmc32_$i -o p -- python $R/tools/gram_probe.py --n 262144 --d 4096 --views 2 --dtype f32 --iters 2 > $O/pmc32_$i.log 2>&1
done
python $R/tools/pmc_extract.py k_gram_f32_fifo $(find /tmp/pmc32_* -name "*results.db") > $O/r05_gram_pmc_raw.md 2>&1
grep -h iter $O/pmc32_1.log | tail -2 >> $O/r05_gram_pmc_raw.md
rm -rf /tmp/pmc32_*
cd $R; timeout 400 python tools/k1_split_probe.py 131072 8192 > $O/r05_k1_split.json 2> $O/k1_split.err; cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/p_loss -o loss -- python $R/tools/loss_profile.py 8192 512 20 > $O/loss_profile.log 2>&1
f=$(find /tmp/p_loss -name "*results.db" | head -1); python $R/tools/rocpd_stats.py $f > $O/r05_loss_c4.md; tail -2 $O/loss_profile.log >> $O/r05_loss_c4.md; rm -rf /tmp/p_loss
prof() { local name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/p_$name -o s -- python $R/tools/solve_probe.py "$@" > $O/solve_${name}.log 2>&1
  f=$(find /tmp/p_$name -name "*results.db" | head -1); python $R/tools/rocpd_stats.py $f k_gemm_f64_big k_gemm_f64_half > $O/r05_solve_${name}.md 2>&1; grep solve $O/solve_${name}.log | tail -3 >> $O/r05_solve_${name}.md; rm -rf /tmp/p_$name; }
prof rcca rcca 4096,4096 64 200000
prof mcca mcca 2048,2048,2048,2048 64 200000
prof gcca gcca 4096,4096,8192 128 60000
cd $R; cat $O/r05_gram_pmc_raw.md; cat $O/r05_k1_split.json; tail -3 $O/k1_split.err; head -14 $O/r05_loss_c4.md; tail -3 $O/r05_solve_rcca.md
