#!/bin/bash
# Round 6: L2 requests and fabric bytes of the split pass with the piece exchange / panel-fast grid (default) against the first form
# (CCZ_SPLIT_XCH=0 CCZ_SPLIT_ORDER=0), n = 262144 rows (as profiles/r06_split_pass_pmc_raw.md).  One counter group per run.  -> stdout
R=$PWD; export TMPDIR=/tmp; cd /tmp
for form in "1 1" "0 0"; do
  set -- $form
  i=0
  for grp in "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES"; do
    i=$((i+1))
    CCZ_SPLIT_XCH=$1 CCZ_SPLIT_ORDER=$2 timeout 200 rocprofv3 --pmc $grp --kernel-trace -d /tmp/pmcx_$i -o p -- python $R/tools/gram_probe.py --n 262144 --d 4096 --views 2 --route bf16x2 --fill latent --iters 2 > /tmp/pmcx_$i.log 2>&1
  done
  echo "== CCZ_SPLIT_XCH=$1 CCZ_SPLIT_ORDER=$2"
  python $R/tools/pmc_extract.py k_split_bf16x2 $(find /tmp/pmcx_* -name "*results.db") | sed 's#/tmp/pmcx_[0-9]*/##'
  rm -rf /tmp/pmcx_*
done
