#!/usr/bin/env python3
"""Dev tool: read a rocprofv3 kernel_trace.csv and print, per kernel name, launch durations bucketed by position in the
run (deciles of the dispatch sequence) plus the largest gaps between consecutive kernels: shows whether a slow phase is
slow kernels or idle time.  usage: trace_tail.py <kernel_trace.csv> [name-substring ...]"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
want = sys.argv[2:] or ["k_tiny_bwd", "k_fproject_bwd", "k_praster_fwd", "k_fproject<", "k_tile_sort", "k_mraster_bwd"]
t0 = int(rows[0]["Start_Timestamp"])
n = len(rows)
print(f"{n} dispatches over {(int(rows[-1]['End_Timestamp']) - t0) / 1e9:.2f} s")
for w in want:
    sel = [(i, r) for i, r in enumerate(rows) if w in r["Kernel_Name"]]
    if not sel:
        continue
    b = defaultdict(list)
    for i, r in sel:
        b[i * 10 // n].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    print(w, " ".join(f"d{k}:n={len(v)} avg={sum(v) / len(v):.1f} max={max(v):.1f}" for k, v in sorted(b.items())))
gaps = []
for a, bb in zip(rows, rows[1:]):
    gaps.append(((int(bb["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e6, (int(a["End_Timestamp"]) - t0) / 1e9,
                 a["Kernel_Name"][:40], bb["Kernel_Name"][:40]))
gaps.sort(reverse=True)
print("largest gaps (ms, at s, after, before):")
for g in gaps[:25]:
    print(f"  {g[0]:9.2f} ms at {g[1]:7.3f} s  {g[2]} -> {g[3]}")
