#!/bin/bash
# Round 6: PMC passes (one counter group per run, never with the trace domains) of the subspace iteration's skinny product
# S X at GCCA's shape (16384^2 x 160) and MCCA's (8192^2 x 80): what holds it at 55 TF of a 78.6 TF pipe.  -> stdout
R=$PWD; export TMPDIR=/tmp; cd /tmp
for shape in "16384 160 16384" "8192 80 8192"; do
  i=0
  for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_MFMA SQ_INSTS_LDS" "SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM"; do
    i=$((i+1))
    timeout 200 rocprofv3 --pmc $grp --kernel-trace -d /tmp/pmck_$i -o p -- python $R/tools/gemm64_probe.py one $shape 0 0 > /tmp/pmck_$i.log 2>&1
  done
  echo "== shape $shape (A not transposed)"
  python $R/tools/pmc_extract.py k_gemm_f64_skinny $(find /tmp/pmck_* -name "*results.db") | sed 's#/tmp/pmck_[0-9]*/##'
  grep -h '^{' /tmp/pmck_1.log | tail -1
  rm -rf /tmp/pmck_*
done
