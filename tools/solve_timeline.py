#!/usr/bin/env python3
"""Kernel timeline of the LAST solve in a rocprofv3 --kernel-trace database of tools/solve_probe.py: every dispatch with its
queue, start, duration and the idle gap of its queue in front of it, plus busy time per queue and of their union -- which
kernels sit on the critical path of the solve stage.

    python tools/solve_timeline.py <results.db> [first-kernel-pattern] > profiles/r06_solve_timeline_rcca.md"""
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
rows = con.execute(f"select start, end, name, {qcol or '0'} from kernels order by start").fetchall()
# the last solve: after the last K1 launch (the moments of the last iteration)
last_k1 = max(i for i, r in enumerate(rows) if "k_gram" in r[2])
sol = [r for r in rows[last_k1 + 1:] if "k_gram" not in r[2]]
t0 = sol[0][0]
queues = sorted({r[3] for r in sol})
qn = {q: i for i, q in enumerate(queues)}
print(f"# solve-stage timeline: {len(sol)} dispatches on {len(queues)} queues, {(max(r[1] for r in sol) - t0) / 1e6:.2f} ms from the first to the last\n")
print("| start us | dur us | gap us | q | kernel |\n|---:|---:|---:|---:|---|")
last_end = {}
busy = {q: 0 for q in queues}
for s, e, name, q in sol:
    gap = (s - last_end[q]) / 1e3 if q in last_end else 0.0
    last_end[q] = e
    busy[q] += e - s
    nm = name.split("(")[0].replace("void ", "").replace("ccz::", "")[:60]
    print(f"| {(s - t0) / 1e3:.1f} | {(e - s) / 1e3:.1f} | {gap:.1f} | {qn[q]} | `{nm}` |")
iv = sorted((r[0], r[1]) for r in sol)
tot, cs, ce = 0, iv[0][0], iv[0][1]
for s, e in iv[1:]:
    if s <= ce:
        ce = max(ce, e)
    else:
        tot += ce - cs
        cs, ce = s, e
tot += ce - cs
print()
for q in queues:
    print(f"* queue {qn[q]}: busy {busy[q] / 1e6:.2f} ms")
print(f"* union of all queues busy {tot / 1e6:.2f} ms; idle (no kernel anywhere) {(max(r[1] for r in sol) - t0 - tot) / 1e6:.2f} ms")
