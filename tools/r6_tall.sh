#!/bin/bash
# Round 6: the fp32 projection kernel in its forms (tools/tall_probe.py: 1 = a row per lane, 2 = whole-line loads with 256 rows
# per workgroup, 3 = with 128 rows) at the metric's view, at a row count that fills the chip in whole rounds (tail effect), at
# d = 512; the GPU tests of transform / score on the default form; PMC passes (fabric bytes, MFMA pipe busy).  -> gpurun_out/tall/
R=$PWD; O=$R/gpurun_out/tall; mkdir -p $O; export TMPDIR=/tmp
export TALL_IMPLS=${TALL_IMPLS:-1,2,3,3,2,1}
python tools/tall_probe.py 1000000 4096 64 > $O/probe.log 2>&1
python tools/tall_probe.py 983040 4096 64 0 >> $O/probe.log 2>&1
python tools/tall_probe.py 4000000 512 64 0 >> $O/probe.log 2>&1
python tools/tall_probe.py 1000000 4096 16 >> $O/probe.log 2>&1
python tools/tall_probe.py 100003 1024 5 >> $O/probe.log 2>&1
cat $O/probe.log
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_estimators.py -q -x -m gpu 2>&1 | tail -5 | tee $O/tests.log
cd /tmp
for impl in ${PMC_IMPLS:-1 2 3}; do
  i=0
  for grp in "FETCH_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAIT_INST_ANY SQ_WAVE_CYCLES"; do
    i=$((i+1))
    TALL_IMPLS=$impl timeout 200 rocprofv3 --pmc $grp --kernel-trace -d /tmp/pmct_$i -o p -- python $R/tools/tall_probe.py 1000000 4096 64 0 > /tmp/pmct_$i.log 2>&1
  done
  echo "== impl $impl" | tee -a $O/pmc.md
  python $R/tools/pmc_extract.py k_gemm_f32_nn_tall $(find /tmp/pmct_* -name "*results.db") | sed 's#/tmp/pmct_[0-9]*/##' | tee -a $O/pmc.md
  rm -rf /tmp/pmct_*
done
