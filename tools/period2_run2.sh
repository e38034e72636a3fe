#!/bin/bash
mkdir -p gpurun_out
python tools/period2_probe2.py plain,manual,pause,spin,plain,gc,fresh,nostore,plain 12 70 > gpurun_out/r3b_p2.log 2>&1
CCZ_TRACE_PHASES=1 python tools/period2_probe2.py plain 10 > gpurun_out/r3b_p2_phases.log 2>&1
grep -h "^##" gpurun_out/r3b_p2.log
grep -h "phases" gpurun_out/r3b_p2_phases.log | tail -10
