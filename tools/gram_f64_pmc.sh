R=$PWD; export TMPDIR=/tmp; cd /tmp
i=0
for grp in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $grp --kernel-trace -d /tmp/pf_$i -o p -- python $R/tools/gram_probe.py --n 131072 --d 4096 --views 2 --dtype f64 --iters 2 > /tmp/pf_$i.log 2>&1
done
python $R/tools/pmc_extract.py k_gram_f64_fifo $(find /tmp/pf_* -name "*results.db") > $R/gpurun_out/f64_pmc_map1.md 2>&1
grep iter /tmp/pf_1.log >> $R/gpurun_out/f64_pmc_map1.md
rm -rf /tmp/pf_*; cat $R/gpurun_out/f64_pmc_map1.md
