#!/bin/bash
# Same-box A/B of bench lines, alternating: REPS=2 bash tools/r6_ab.sh "<flags A>" "<flags B>" ...   (summaries under gpurun_out/r6ab)
set -e
R=$GRAFT_REPO_ROOT
S=$R/gpurun_out/${OUT:-r6ab}
mkdir -p $S
cd $R
B="python3 bench.py --steps ${STEPS:-8} --warmup 3 --no-parity --no-cpu-baseline --no-b1"
for rep in $(seq 1 ${REPS:-2}); do
  i=0
  for flags in "$@"; do
    i=$((i+1))
    timeout -k 10 300 $B $flags > $S/ab_${i}_$rep.json 2> $S/ab_${i}_$rep.err
    python3 -c "import json; d=json.load(open('$S/ab_${i}_$rep.json')); print('[$flags]', round(d['value'],2), 'frames/s', round(d['ms_per_step'],1), 'ms  frac', d['roofline']['frac'])" | tee -a $S/summary.txt
  done
done
