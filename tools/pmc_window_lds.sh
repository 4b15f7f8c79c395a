#!/bin/bash
# Where do the LDS bank-conflict cycles of the window attention come from? One rocprofv3 --pmc pass per ablated build
# (K reads / V^T reads / rel-pos scratch / staging DMA left out), the counters of window_attn_kernel averaged per launch.
#       gpurun --timeout 900 -- 'bash tools/pmc_window_lds.sh'          (builds the variants first, on the box)
set -e
R=$GRAFT_REPO_ROOT
S=$R/gpurun_out/r6w
mkdir -p $S
cd $R
bash tools/build_window_variant.sh base
bash tools/build_window_variant.sh nokread -DHAFF_WIN_NOKREAD
bash tools/build_window_variant.sh novread -DHAFF_WIN_NOVREAD
bash tools/build_window_variant.sh noscr -DHAFF_WIN_NOSCR
bash tools/build_window_variant.sh nodma -DHAFF_WIN_NODMA
bash tools/build_window_variant.sh nokv -DHAFF_WIN_NOKREAD -DHAFF_WIN_NOVREAD
cd /tmp && export TMPDIR=/tmp
for v in base nokread novread noscr nodma nokv; do
  VARIANT=$v rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d /tmp/r6w/pmc_$v -o x -- python3 $R/tools/window_variant.py > $S/run_$v.txt 2>&1
done
python3 - <<PY | tee $S/summary.txt
import csv, glob, os
from collections import defaultdict
for v in ("base","nokread","novread","noscr","nodma","nokv"):
    acc=defaultdict(float); n=defaultdict(int)
    for f in glob.glob(f"/tmp/r6w/pmc_{v}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "window_attn_kernel" in r["Kernel_Name"]:
                acc[r["Counter_Name"]]+=float(r["Counter_Value"]); n[r["Counter_Name"]]+=1
    print("%-8s" % v, {k: round(acc[k]/max(n[k],1)) for k in sorted(acc)}, "launches", max(n.values()) if n else 0)
PY
grep -h "us per" $S/run_*.txt | tee -a $S/summary.txt
