#!/bin/bash
# Round 6: the split pass with the panels fastest in the grid (CCZ_SPLIT_ORDER=1: the workgroups that run together read whole
# rows) against the row blocks fastest (default): k1_stages_ms.split of the short bench, alternating.  -> stdout
for o in 1 0 1 0 1 0; do
  echo "== CCZ_SPLIT_ORDER=$o"
  CCZ_SPLIT_ORDER=$o python bench.py --steps 10 --warmup 2 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('ms_per_step', round(d['ms_per_step'],2), 'k1', round(r['k1_ms'],2), r['k1_stages_ms'], 'solve', round(d['phases_ms']['solve'],2))"
done
