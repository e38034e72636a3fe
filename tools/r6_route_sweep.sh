#!/bin/bash
# Round 6: both K1 routes over a sweep of row counts at two widths (error against float64 moments of the same rows, stage times):
# where the split route starts to be at least as accurate as the fp32 kernel, and where it starts to pay.  -> stdout
for d in 1024 4096; do
  for n in 4096 8192 16384 32768 65536 131072 262144; do
    echo "== two views n=$n d=$d"
    timeout 300 python tools/k1_route_check.py big $n $d 2>&1 | grep '^{' | python -c "
import sys, json
rows = [json.loads(l) for l in sys.stdin]
best = {}
for r in rows:
    k = r['route']
    if k not in best or r['gram_ms'] < best[k]['gram_ms']:
        best[k] = r
for k, r in best.items():
    print(k, 'k1_ms', r['gram_ms'], 'cov_max', r.get('cov_max'), 'raw_max', r.get('raw_max'), *(('mfma_ms', r['mfma_ms'], 'exec_bf16_TF', r['exec_bf16_TF']) if k == 'bf16x2' else ()))
"
  done
done
