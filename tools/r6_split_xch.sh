#!/bin/bash
# Round 6: the split pass with the wave-private piece exchange in front of its stores (default) against the direct 16-byte
# stores (CCZ_SPLIT_XCH=0): the split-route tests, then the short bench (k1_stages_ms.split) both ways.  -> stdout
python -m pytest tests/test_gpu_k1_split.py -q -x -m gpu 2>&1 | tail -3
for x in 1 0 1 0; do
  echo "== CCZ_SPLIT_XCH=$x"
  CCZ_SPLIT_XCH=$x python bench.py --steps 5 --warmup 2 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('ms_per_step', round(d['ms_per_step'],2), 'k1', round(r['k1_ms'],2), r['k1_stages_ms'], 'solve', round(d['phases_ms']['solve'],2))"
done
