R=$PWD; O=$R/gpurun_out/tall; mkdir -p $O
TALL_IMPLS=1,3w,3,3,3w python tools/tall_probe.py 1000000 4096 16 > $O/probe_nj.log 2>&1
TALL_IMPLS=1,3w,3,3,3w python tools/tall_probe.py 1000000 4096 32 >> $O/probe_nj.log 2>&1
TALL_IMPLS=1,3w,3,3,3w python tools/tall_probe.py 100003 1024 5 >> $O/probe_nj.log 2>&1
TALL_IMPLS=1,3w,3,3,3w python tools/tall_probe.py 4000000 512 8 >> $O/probe_nj.log 2>&1
grep -v amdgpu.ids $O/probe_nj.log
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_estimators.py -q -x -m gpu 2>&1 | tail -3
