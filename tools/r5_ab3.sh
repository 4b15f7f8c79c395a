#!/bin/bash
# same-box A/B of bench lines: name=args pairs, alternated REPS times
mkdir -p gpurun_out/r5c
REPS=${REPS:-2}
for rep in $(seq 1 $REPS); do
  i=0
  for a in "$@"; do
    i=$((i+1))
    python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-parity --no-b1 $a > gpurun_out/r5c/ab3_${i}_$rep.json 2> gpurun_out/r5c/ab3_${i}_$rep.err || exit 1
    python3 -c "
import json
d=json.load(open('gpurun_out/r5c/ab3_${i}_$rep.json')); print('[$a]', round(d['value'],2), round(d['ms_per_step'],1), d['config'].get('sam_chunk_workgroup_caps'))"
  done
done
