#!/bin/bash
mkdir -p gpurun_out
SECONDS=0; python bench.py > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err; echo "bench.py wall: $SECONDS s"; tail -c 300 gpurun_out/final_bench.json; grep -i "PARITY\|Error\|Traceback" gpurun_out/final_bench.err | head
