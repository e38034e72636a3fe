#!/bin/bash
# PMC passes over the blocked-Jacobi kernels at d = 4096 (one counter group per pass, kernel trace only):
# MFMA pipe occupancy and memory-side traffic of k_bj_apply / k_bj_inner2 -> gpurun_out/evd_pmc.md
R=$PWD; export TMPDIR=/tmp; cd /tmp
i=0
for grp in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 250 rocprofv3 --pmc $grp --kernel-trace -d /tmp/pe_$i -o p -- python $R/tools/evd_probe.py syev ${1:-4096} 0 > /tmp/pe_$i.log 2>&1
done
{ python $R/tools/pmc_extract.py k_bj_apply $(find /tmp/pe_* -name "*results.db"); python $R/tools/pmc_extract.py k_bj_inner2 $(find /tmp/pe_* -name "*results.db"); } > $R/gpurun_out/evd_pmc.md 2>&1
rm -rf /tmp/pe_*; cat $R/gpurun_out/evd_pmc.md
