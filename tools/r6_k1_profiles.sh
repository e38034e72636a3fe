#!/bin/bash
# Round-6 evidence for the split-bf16 K1 route: DVFS check (zero-filled vs random operands), kernel statistics and one PMC
# pass per counter group (never combined with trace domains).  -> gpurun_out/r6p/
R=$PWD; O=$R/gpurun_out/r6p; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
N=${1:-262144}
for fill in randn zeros latent; do
  timeout 200 python $R/tools/gram_probe.py --n $N --d 4096 --views 2 --route bf16x2 --fill $fill --iters 3 > $O/fill_$fill.log 2>&1
done
timeout 200 python $R/tools/gram_probe.py --n $N --d 4096 --views 2 --route fp32 --fill latent --iters 2 > $O/fill_latent_fp32.log 2>&1
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $grp --kernel-trace -d /tmp/pmcs_$i -o p -- python $R/tools/gram_probe.py --n $N --d 4096 --views 2 --route bf16x2 --fill latent --iters 2 > $O/pmcs_$i.log 2>&1
done
for k in k_gram_bf16x2 k_split_bf16x2 k_split_reduce; do
  python $R/tools/pmc_extract.py $k $(find /tmp/pmcs_* -name "*results.db") > $O/r06_${k}_pmc_raw.md 2>&1
done
grep -h "iter" $O/pmcs_1.log | tail -4 >> $O/r06_k_gram_bf16x2_pmc_raw.md
rm -rf /tmp/pmcs_*
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/p_k1 -o k1 -- python $R/tools/gram_probe.py --n $N --d 4096 --views 2 --route bf16x2 --fill latent --iters 3 > $O/k1_stats.log 2>&1
f=$(find /tmp/p_k1 -name "*results.db" | head -1); python $R/tools/rocpd_stats.py $f > $O/r06_k1_kernel_stats.md 2>&1; rm -rf /tmp/p_k1
cd $R; for f in $O/fill_*.log; do tail -n 3 $f; done; cat $O/r06_k_gram_bf16x2_pmc_raw.md; head -20 $O/r06_k1_kernel_stats.md
