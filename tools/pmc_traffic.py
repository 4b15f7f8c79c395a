#!/usr/bin/env python3
"""Summarise the two rocprofv3 --pmc passes of bench.py (FETCH_SIZE, WRITE_SIZE — they do not fit one pass) into the
per-launch HBM traffic of the dominant kernel family, written to profiles/pmc_gemm_traffic.json for bench.py's
`roofline.traffic`. Units and correction per MI355X_MICROARCH.md: both counters are KiB; on gfx950 FETCH_SIZE counts
128-B requests as 64 B, so it is doubled; WRITE_SIZE is exact for 16-B-per-lane stores.
usage: pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <config> <batch> [out.json]"""
import collections
import csv
import json
import os
import sys


def per_kernel(path, counter):
    tot = collections.defaultdict(lambda: [0, 0.0])
    with open(path) as fh:
        for r in csv.DictReader(fh):
            if r["Counter_Name"] != counter:
                continue
            name = r["Kernel_Name"]
            # the family bench.py meters as the dominant kernel: every tile launch with more than 64 rows = the 256 x 256 and
            # 192 x 256 instances and the 128 x 128 instance with a bf16 output; the 128 x 128 instance that writes fp32 partial
            # tiles is the split-K half of the 33..64-row decode products (the weight-streaming family), listed on its own
            if "gemm_bf16_kernel" in name:
                key = "gemm_bf16_kernel (split-K partials, decode)" if "gemm_bf16_kernel<128, 128, 2, 2, true" in name else "gemm_bf16_kernel"
            else:
                key = name.replace("(anonymous namespace)::", "")[:60]
            tot[key][0] += 1
            tot[key][1] += float(r["Counter_Value"])
    return tot


def per_instance(path, counter):
    """gemm_bf16_kernel launches by template instance (tile, epilogue flags) and grid: {key: [launches, counter sum]}"""
    import re
    tot = collections.defaultdict(lambda: [0, 0.0])
    with open(path) as fh:
        for r in csv.DictReader(fh):
            if r["Counter_Name"] != counter or "gemm_bf16_kernel" not in r["Kernel_Name"]:
                continue
            m = re.search(r"gemm_bf16_kernel<([^>]*)>", r["Kernel_Name"])
            key = (m.group(1) if m else "?") + " grid " + r.get("Grid_Size", r.get("Grid_Size_X", "?"))
            tot[key][0] += 1
            tot[key][1] += float(r["Counter_Value"])
    return tot


def main():
    fpath, wpath, config, batch = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
    out = sys.argv[5] if len(sys.argv) > 5 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                            "profiles", "pmc_gemm_traffic.json")
    f, w = per_kernel(fpath, "FETCH_SIZE"), per_kernel(wpath, "WRITE_SIZE")
    table = {}
    for k in f:
        n = f[k][0]
        fetch = 2.0 * 1024.0 * f[k][1] / n
        write = 1024.0 * w[k][1] / w[k][0] if k in w else 0.0
        table[k] = {"launches": n, "fetch_bytes_per_launch": fetch, "write_bytes_per_launch": write}
    g = table["gemm_bf16_kernel"]
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import haff  # noqa: F401
    from haff import lib as hlib
    res = {"config": config, "batch": batch, "kernel": "gemm_bf16_kernel",
           "library_source_sha16": hlib.source_hash(),   # bench.py reports traffic: null when the tree's sources differ
           "launches_profiled": g["launches"],
           "hbm_bytes_per_launch": g["fetch_bytes_per_launch"] + g["write_bytes_per_launch"],
           "fetch_bytes_per_launch": g["fetch_bytes_per_launch"], "write_bytes_per_launch": g["write_bytes_per_launch"],
           "collected": "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE --kernel-trace -- python3 bench.py --batch %d --steps 1 "
                        "--warmup 1 --no-cpu-baseline --no-b1; FETCH_SIZE x2 (gfx950)" % batch,
           "other_kernels": {k: v for k, v in sorted(table.items(), key=lambda kv: -kv[1]["fetch_bytes_per_launch"] * kv[1]["launches"])[:10]
                             if k != "gemm_bf16_kernel"}}
    fi, wi = per_instance(fpath, "FETCH_SIZE"), per_instance(wpath, "WRITE_SIZE")
    res["by_instance"] = {k: {"launches": v[0], "fetch_bytes_per_launch": 2.0 * 1024.0 * v[1] / v[0],
                              "write_bytes_per_launch": (1024.0 * wi[k][1] / wi[k][0]) if k in wi else None}
                          for k, v in sorted(fi.items(), key=lambda kv: -kv[1][1])[:24]}
    with open(out, "w") as fh:
        json.dump(res, fh, indent=1)
    print(json.dumps({k: res[k] for k in ("launches_profiled", "hbm_bytes_per_launch", "fetch_bytes_per_launch", "write_bytes_per_launch")}))


if __name__ == "__main__":
    main()
