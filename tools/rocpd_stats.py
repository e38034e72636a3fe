#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd SQLite database (``*_results.db``) into a kernel-stats table.

    python tools/rocpd_stats.py gpurun_out/prof_r1/ns_results.db > profiles/r01_ns_kernel_stats.md

Equivalent of ``rocprofv3 --stats`` CSV: per kernel calls, total / average / min / max duration
and share of GPU kernel time.
"""

import re
import sqlite3
import sys


def short(name):
    name = name.replace("(anonymous namespace)::", "")
    name = re.sub(r"\(.*$", "", name)
    name = re.sub(r"^void ", "", name)
    return name[:90]


def by_grid(con, pattern):
    """Per launch-shape breakdown of the kernels matching ``pattern`` (same kernel, different problem sizes)."""
    rows = con.execute(
        "select name, grid_x, grid_y, grid_z, count(*), avg(duration), min(duration), max(duration) from kernels "
        "where name like ? group by name, grid_x, grid_y, grid_z order by avg(duration) desc", (f"%{pattern}%",)).fetchall()
    print(f"\nlaunch shapes of `*{pattern}*`:\n")
    print("| kernel | grid (threads) | calls | avg us | min us | max us |")
    print("|---|---|---:|---:|---:|---:|")
    for name, gx, gy, gz, calls, avg, mn, mx in rows:
        print(f"| `{short(name)}` | {gx} x {gy} x {gz} | {calls} | {avg / 1e3:.1f} | {mn / 1e3:.1f} | {mx / 1e3:.1f} |")


def main(path, patterns=()):
    con = sqlite3.connect(path)
    rows = con.execute(
        "select name, count(*), sum(duration), avg(duration), min(duration), max(duration) "
        "from kernels group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    print(f"source: {path}")
    print(f"total GPU kernel time: {total / 1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches\n")
    print("| kernel | calls | total ms | avg us | min us | max us | % |")
    print("|---|---:|---:|---:|---:|---:|---:|")
    for name, calls, tot, avg, mn, mx in rows:
        print(f"| `{short(name)}` | {calls} | {tot / 1e6:.3f} | {avg / 1e3:.1f} | {mn / 1e3:.1f} | {mx / 1e3:.1f} | {100 * tot / total:.2f} |")
    for pat in patterns:
        by_grid(con, pat)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2:])
