#!/usr/bin/env python3
"""profiles/r06_gram_traffic.json from the raw PMC table of k_gram_bf16x2 (tools/pmc_extract.py output): fabric bytes per
launch = FETCH_SIZE x 1024 x 2 (gfx950 correction, MI355X_MICROARCH.md "HBM") + WRITE_SIZE x 1024, per row of the launch.

    python tools/pmc_to_traffic.py gpurun_out/r6p/r06_k_gram_bf16x2_pmc_raw.md 262144 8192 > profiles/r06_gram_traffic.json
"""
import json
import re
import sys

raw, n, D = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
vals = {}
for line in open(raw):
    m = re.match(r"\|\s*(\w+)\s*\|\s*\d+\s*\|\s*([0-9.e+]+)\s*\|", line)
    if m:
        vals[m.group(1)] = float(m.group(2))
fetch = vals["FETCH_SIZE"] * 1024 * 2
write = vals["WRITE_SIZE"] * 1024
out = {"kernel": "k_gram_bf16x2", "dtype": "bf16x2", "n": n, "D": D, "fetch_bytes_per_launch": fetch, "write_bytes_per_launch": write,
       "bytes_per_row": (fetch + write) / n, "algorithmic_bytes_per_launch": float(n) * D * 4,
       "mfma_pipe_busy": vals.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (1024 * vals["GRBM_GUI_ACTIVE"] / 8) if vals.get("GRBM_GUI_ACTIVE") else None,
       "l2_hit_rate": vals["TCC_HIT_sum"] / (vals["TCC_HIT_sum"] + vals["TCC_MISS_sum"]) if "TCC_HIT_sum" in vals else None,
       "source": "profiles/r06_gram_bf16x2_pmc_raw.md", "note": "the MFMA kernel of the split route only (the split pass reads the fp32 rows once and "
       "writes the planes once; the reduce reads the partial tiles once): see profiles/r06_gram_pmc.md"}
print(json.dumps(out, indent=1))
