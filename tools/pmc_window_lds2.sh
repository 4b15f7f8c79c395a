#!/bin/bash
# Round 6: candidates against the two sources tools/pmc_window_lds.sh named (rel-pos scratch 51 %, last K k-step 34 % of the conflict cycles)
set -e
R=$GRAFT_REPO_ROOT
S=$R/gpurun_out/r6w2
mkdir -p $S
cd $R
bash tools/build_window_variant.sh old -DHAFF_WIN_KLAST_OLD
bash tools/build_window_variant.sh base
for rs in 40 44 52 68; do bash tools/build_window_variant.sh rs$rs -DHAFF_WIN_RS=$rs; done
cd /tmp && export TMPDIR=/tmp
for v in old base rs40 rs44 rs52 rs68; do
  VARIANT=$v rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d /tmp/r6w2/pmc_$v -o x -- python3 $R/tools/window_variant.py > $S/run_$v.txt 2>&1
done
python3 - <<PY | tee $S/summary.txt
import csv, glob, os
from collections import defaultdict
for v in ("old","base","rs40","rs44","rs52","rs68"):
    acc=defaultdict(float); n=defaultdict(int)
    for f in glob.glob(f"/tmp/r6w2/pmc_{v}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "window_attn_kernel" in r["Kernel_Name"]:
                acc[r["Counter_Name"]]+=float(r["Counter_Value"]); n[r["Counter_Name"]]+=1
    print("%-8s" % v, {k: round(acc[k]/max(n[k],1)) for k in sorted(acc)}, "launches", max(n.values()) if n else 0)
PY
cd $R
for v in old base rs40 rs44 rs52 rs68 old base; do VARIANT=$v python3 tools/window_variant.py 2>/dev/null | tee -a $S/summary.txt; done
