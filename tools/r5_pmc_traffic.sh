#!/bin/bash
# The two --pmc passes behind roofline.traffic (FETCH_SIZE, WRITE_SIZE: separate runs), summarised into profiles-ready json,
# then the default bench line that reads it.     gpurun --timeout 900 -- 'bash tools/r5_pmc_traffic.sh [extra bench args]'
set -e
R=$GRAFT_REPO_ROOT; S=$R/gpurun_out/r5e; O=/tmp/r5pmc; mkdir -p $S $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o x -- python3 $R/bench.py --batch 64 --steps 1 --warmup 1 --no-cpu-baseline --no-b1 --no-parity $@ > $S/pmc_fetch.txt 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o x -- python3 $R/bench.py --batch 64 --steps 1 --warmup 1 --no-cpu-baseline --no-b1 --no-parity $@ > $S/pmc_write.txt 2>&1
python3 $R/tools/pmc_traffic.py $(find $O/pmc_fetch -name '*counter_collection.csv' | head -1) $(find $O/pmc_write -name '*counter_collection.csv' | head -1) 2HandedAfforder-7B 64 $S/pmc_gemm_traffic.json
head -c 600 $(find $O/pmc_fetch -name '*counter_collection.csv' | head -1)
echo done
