#!/usr/bin/env python3
"""Per-kernel durations AND the gaps between consecutive kernels of KV-cached decode steps, from a rocprofv3 --kernel-trace csv.
usage: python3 tools/decode_trace.py <kernel_trace.csv> [n_layers]
Takes the LAST run of n_layers x (layer pattern) in the trace (the timed steps replay one hipGraph) and prints, per position of the
layer's launch pattern, the mean duration, the mean gap to the next launch and the share of the layer time."""
import csv
import sys
from collections import defaultdict


def short(name):
    for key in ("gemm_skinny_kernel", "attn_decode_kernel", "skinny_reduce_kernel", "decode_chain_kernel", "norm_row", "norm_rows",
                "gemm_bf16_kernel", "argmax", "decode_book", "embed", "row_stats"):
        if key in name:
            i = name.find(key)
            return name[i:i + 48].split("(")[0]
    return name[:48]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    ev = [(short(r["Kernel_Name"]), int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
    # the decode steps: find the last 4000 launches, aggregate by kernel name
    tail = ev[-int(sys.argv[2]) if len(sys.argv) > 2 else -2000:]
    dur, gap, cnt = defaultdict(float), defaultdict(float), defaultdict(int)
    for i, (n, s, e) in enumerate(tail[:-1]):
        dur[n] += (e - s) / 1e3
        gap[n] += max(0, tail[i + 1][1] - e) / 1e3
        cnt[n] += 1
    span = (tail[-1][2] - tail[0][1]) / 1e3
    print("span of the last %d launches: %.1f us" % (len(tail), span))
    print("%-50s %7s %9s %9s %8s" % ("kernel", "calls", "avg us", "gap us", "share"))
    for n in sorted(dur, key=lambda k: -dur[k]):
        print("%-50s %7d %9.2f %9.2f %7.1f%%" % (n, cnt[n], dur[n] / cnt[n], gap[n] / cnt[n], 100 * (dur[n] + gap[n]) / span))
    print("sum of durations %.1f us, sum of gaps %.1f us" % (sum(dur.values()), sum(gap.values())))


if __name__ == "__main__":
    main()
