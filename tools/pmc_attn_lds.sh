set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r4i
for v in base novread nokread nodma; do
  VARIANTS=$v B=8 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $R/gpurun_out/r4i/pmc_$v -o x -- python3 $R/tools/attn_variant.py > $R/gpurun_out/r4i/run_$v.txt 2>&1
done
python3 - <<PY
import csv, glob, os
from collections import defaultdict
R=os.environ["GRAFT_REPO_ROOT"]
for v in ("base","novread","nokread","nodma"):
    acc=defaultdict(float); n=defaultdict(int)
    for f in glob.glob(f"{R}/gpurun_out/r4i/pmc_{v}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "attn_global_pp" in r["Kernel_Name"]:
                acc[r["Counter_Name"]]+=float(r["Counter_Value"]); n[r["Counter_Name"]]+=1
    print(v, {k: round(acc[k]/max(n[k],1)) for k in sorted(acc)}, "launches", max(n.values()) if n else 0)
PY
grep -h "us " $R/gpurun_out/r4i/run_*.txt
