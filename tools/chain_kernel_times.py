#!/usr/bin/env python3
"""Durations of every decode_chain_kernel dispatch in a rocprofv3 --kernel-trace csv, in order, grouped by grid size (= batch size)."""
import csv
import sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "decode_chain_kernel" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
out = []
for r in rows:
    g = r.get("Grid_Size") or r.get("Grid_Size_X") or "?"
    out.append((g, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
prev, acc = None, []
for g, d in out + [(None, 0)]:
    if g != prev and acc:
        print(f"grid {prev}: {len(acc)} dispatches, us: first {acc[0]:.0f}, median {sorted(acc)[len(acc) // 2]:.0f}, min {min(acc):.0f}, max {max(acc):.0f}")
        acc = []
    prev = g
    acc.append(d)
