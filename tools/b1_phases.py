#!/usr/bin/env python3
"""Phase breakdown of one batch-1 evaluate() from a rocprofv3 kernel trace CSV (tools/b1_run.py --single-stream)."""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"at::native::", "", n)
    return n[:64]


starts = [i for i, r in enumerate(rows) if "resample_w" in r["Kernel_Name"]]
a, b = starts[-2], starts[-1]
ev = rows[a:b]
t0 = int(ev[0]["Start_Timestamp"])
T = lambda r: ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3)
marks = {}
for i, r in enumerate(ev):
    k = short(r["Kernel_Name"])
    for name, pat in (("clip", "patchify_nchw"), ("prefill", "embed_splice"), ("decode", "argmax_rows"), ("tail", "bfloat16tofloat32")):
        if name not in marks and pat in k:
            marks[name] = i
order = [("sam+ingest", 0)] + sorted(marks.items(), key=lambda x: x[1])
order.append(("end", len(ev)))
print(f"evaluate span {T(ev[-1])[1] / 1e3:.2f} ms, {len(ev)} kernels")
for (name, i0), (_, i1) in zip(order, order[1:]):
    seg = ev[i0:i1]
    span = T(seg[-1])[1] - T(seg[0])[0]
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in seg:
        s, e = T(r)
        agg[short(r["Kernel_Name"])][0] += 1
        agg[short(r["Kernel_Name"])][1] += e - s
    print(f"== {name}: {span / 1e3:.2f} ms, {len(seg)} kernels")
    for k, (n, t) in sorted(agg.items(), key=lambda x: -x[1][1])[:8]:
        print(f"   {t:8.1f} us {n:4d} x {t / n:7.2f}  {k}")
