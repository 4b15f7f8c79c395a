"""Idle time of the GPU in a rocprofv3 --kernel-trace CSV: merges the kernels' [start, end] intervals (all streams), lists the
largest gaps with the kernels on either side, and the total idle time inside [t0, t1] = the last `--last` fraction of the trace
(the timed steps). python tools/gap_report.py trace.csv [--last 0.5] [--min-us 30]"""
import csv
import sys

path = sys.argv[1]
last = float(sys.argv[sys.argv.index("--last") + 1]) if "--last" in sys.argv else 0.5
min_us = float(sys.argv[sys.argv.index("--min-us") + 1]) if "--min-us" in sys.argv else 30.0
rows = []
with open(path) as fh:
    rd = csv.DictReader(fh)
    for r in rd:
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
t_begin, t_end = rows[0][0], max(r[1] for r in rows)
t0 = t_end - (t_end - t_begin) * last
sel = [r for r in rows if r[0] >= t0]
gaps = []
cur_end, cur_name = sel[0][1], sel[0][2]
busy = 0
seg_start = sel[0][0]
for s, e, n in sel[1:]:
    if s > cur_end:
        gaps.append((s - cur_end, cur_name, n, cur_end))
        busy += cur_end - seg_start
        seg_start = s
        cur_end, cur_name = e, n
    elif e > cur_end:
        cur_end, cur_name = e, n
busy += cur_end - seg_start
span = cur_end - sel[0][0]
idle = sum(g[0] for g in gaps)
print(f"window {span / 1e6:.1f} ms, busy {busy / 1e6:.1f} ms, idle {idle / 1e6:.1f} ms ({100 * idle / span:.1f} %), {len(sel)} kernels")
big = [g for g in gaps if g[0] >= min_us * 1e3]
print(f"gaps >= {min_us:.0f} us: {len(big)}, total {sum(g[0] for g in big) / 1e6:.1f} ms; small gaps total {(idle - sum(g[0] for g in big)) / 1e6:.1f} ms")
for g in sorted(big, reverse=True)[:25]:
    print(f"  {g[0] / 1e3:9.1f} us  after {g[1][:70]:70s} before {g[2][:70]}")
