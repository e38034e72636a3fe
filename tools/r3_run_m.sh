#!/bin/bash
python - <<'PY'
import sys, json; sys.path.insert(0,'.')
import bench
from cca_zoo_amd import _backend
out = bench.config_extras(_backend.default_handle().device_info(), gates=True)
for k,v in out.items(): print(k, v.get('fit_ms'), v.get('solve_ms'), v.get('solve_ms_runs'), v.get('gram_frac_of_peak'), v.get('parity_gate',{}).get('ok'))
PY
