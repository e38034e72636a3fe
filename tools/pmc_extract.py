#!/usr/bin/env python3
"""Per-dispatch PMC counter values of one kernel from rocprofv3 rocpd databases.

    python tools/pmc_extract.py k_gram_f32_fifo gpurun_out/pmc3_*/p_results.db

One ``rocprofv3 --pmc <counters> --kernel-trace`` pass per counter group (never combined with the
sys/hip/hsa trace domains); values are summed over the counter's instances (XCDs / channels) and
averaged over the kernel's dispatches, the largest grid only.
"""
import sqlite3
import sys


def main(pattern, paths):
    print(f"kernel pattern: *{pattern}*\n")
    print("| counter | dispatches | per dispatch | source |")
    print("|---|---:|---:|---|")
    for path in paths:
        con = sqlite3.connect(path)
        gmax = con.execute("select max(grid_size) from counters_collection where kernel_name like ?", (f"%{pattern}%",)).fetchone()[0]
        if gmax is None:
            continue
        rows = con.execute(
            "select counter_name, count(distinct dispatch_id), sum(value) from counters_collection "
            "where kernel_name like ? and grid_size = ? group by counter_name", (f"%{pattern}%", gmax)).fetchall()
        for name, nd, tot in rows:
            print(f"| {name} | {nd} | {tot / nd:.6g} | {path} |")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2:])
