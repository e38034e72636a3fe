#!/usr/bin/env python3
"""Timeline of ONE CCA.fit on pinned host views from a rocprofv3 database taken with --kernel-trace --memory-copy-trace
(tools/r5_host_timeline.sh): every host -> device copy >= 1 MiB and every K1 / column-sum launch of the LAST fit, on one
time axis, plus the busy time of the copy engine and of the compute queue and their overlap.

    python tools/host_timeline.py gpurun_out/hostfit/h_results.db > profiles/r05_host_fit_timeline.md"""
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
copies = con.execute("select start, end, size, name from memory_copies where size >= 1048576 order by start").fetchall()
kern = con.execute("select start, end, name from kernels where name like '%k_gram%' or name like '%k_colsum%' order by start").fetchall()
# the last fit = the last run of big copies separated from the previous one by > 15 ms
groups, cur = [], []
for c in copies:
    if cur and c[0] - cur[-1][1] > 15_000_000:
        groups.append(cur)
        cur = []
    cur.append(c)
groups.append(cur)
last = groups[-1]
t0, t1 = last[0][0], max(c[1] for c in last)
ks = [k for k in kern if k[0] >= t0 - 1_000_000 and k[0] <= t1 + 50_000_000]
t_end = max([t1] + [k[1] for k in ks])
ev = [(c[0], c[1], "H2D copy", f"{c[2] / 2**20:.0f} MiB", c[2]) for c in last] + \
     [(k[0], k[1], "kernel", k[2].split("(")[0].replace("void ", "").replace("ccz::", "")[:40], 0) for k in ks]
ev.sort()
print("# Round 5 -- `CCA(64).fit` on pinned HOST views, 262 144 rows x 2 x 4096 fp32 (8.6 GB): copy / kernel timeline of one fit\n")
print("`rocprofv3 --kernel-trace --memory-copy-trace -- python tools/host_fit_probe2.py` (`tools/r5_host_timeline.sh`), last of five fits;")
print("times in ms from the first copy's start.  Behind `bench.py`'s `extra.host_inputs` (VERDICT r4 item 7).\n")
print("| start | end | ms | what | |\n|---:|---:|---:|---|---|")
for s, e, kind, what, _ in ev:
    print(f"| {(s - t0) / 1e6:.2f} | {(e - t0) / 1e6:.2f} | {(e - s) / 1e6:.2f} | {kind} | {what} |")


def union(iv):
    iv = sorted(iv)
    out, tot = [], 0
    for s, e in iv:
        if out and s <= out[-1][1]:
            out[-1][1] = max(out[-1][1], e)
        else:
            out.append([s, e])
    return out, sum(e - s for s, e in out)


cu, cbusy = union([(c[0], c[1]) for c in last])
ku, kbusy = union([(k[0], k[1]) for k in ks])
ov = 0
for a in cu:
    for b in ku:
        ov += max(0, min(a[1], b[1]) - max(a[0], b[0]))
nbytes = sum(c[2] for c in last)
span = t_end - t0
print(f"\n* bytes copied: {nbytes / 1e9:.2f} GB in {len(last)} copies; copy engine busy {cbusy / 1e6:.1f} ms "
      f"= {nbytes / cbusy:.1f} GB/s while a copy is running; first copy start -> last copy end {(t1 - t0) / 1e6:.1f} ms "
      f"= {nbytes / (t1 - t0):.1f} GB/s")
print(f"* K1 + column sums busy {kbusy / 1e6:.1f} ms, of which {ov / 1e6:.1f} ms under a copy; after the last copy ends: "
      f"{(t_end - t1) / 1e6:.1f} ms of kernels (the drain)")
print(f"* first copy start -> last K1 end: {span / 1e6:.1f} ms = {nbytes / span:.1f} GB/s for the moments phase")
gaps = [(cu[i + 1][0] - cu[i][1]) / 1e6 for i in range(len(cu) - 1)]
print(f"* gaps between consecutive copies (ms): {[round(g, 2) for g in gaps]}")
