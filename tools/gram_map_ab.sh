#!/bin/bash
# K1 work mapping A/B: CCZ_GRAM_MAP=0 (per-XCD tile slices, round 1/2) vs 1 (chunk-per-XCD): time at the metric shape,
# then fabric reads / L2 hits at n = 262144 (one --pmc group per pass).  -> gpurun_out/gram_map_ab.log
R=$PWD; mkdir -p gpurun_out; export TMPDIR=/tmp; log=$R/gpurun_out/gram_map_ab.log; : > $log
for m in 1 0 1 0; do
  echo "== CCZ_GRAM_MAP=$m time, n=1e6 f32 / n=5e5 f64" >> $log
  CCZ_GRAM_MAP=$m python tools/gram_probe.py --n 1000000 --d 4096 --views 2 --dtype f32 --iters 3 2>&1 | grep iter >> $log
  CCZ_GRAM_MAP=$m python tools/gram_probe.py --n 500000 --d 4096 --views 2 --dtype f64 --iters 2 2>&1 | grep iter >> $log
done
cd /tmp
for m in 1; do
  i=0
  for grp in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum" "WRITE_SIZE"; do
    i=$((i+1))
    CCZ_GRAM_MAP=$m timeout 200 rocprofv3 --pmc $grp --kernel-trace -d /tmp/pm${m}_$i -o p -- python $R/tools/gram_probe.py --n 262144 --d 4096 --views 2 --dtype f32 --iters 2 > /tmp/pm${m}_$i.log 2>&1
  done
  echo "== CCZ_GRAM_MAP=$m counters (n=262144, f32)" >> $log
  python $R/tools/pmc_extract.py k_gram_f32_fifo $(find /tmp/pm${m}_* -name "*results.db") >> $log 2>&1
  rm -rf /tmp/pm${m}_*
done
cd $R; cat $log
