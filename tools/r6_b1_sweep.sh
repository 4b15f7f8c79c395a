#!/bin/bash
# Round 6: one frame per step (BASELINE configs[1]) — where the encoder runs and on how many CUs. One box, alternating.
set -e
R=$GRAFT_REPO_ROOT
S=$R/gpurun_out/r6b1
mkdir -p $S
cd $R
B="python3 bench.py --batch 1 --steps 30 --warmup 5 --no-parity --no-cpu-baseline --no-b1"
for rep in 1 2; do
  i=0
  for flags in "" "--sam-beside-decode off" "--sam-caps 128 --sam-waits-for-prefill on" "--sam-caps 64 --sam-waits-for-prefill on" "--sam-caps 192 --sam-waits-for-prefill on" "--sam-caps 128 --sam-waits-for-prefill off" "--single-stream"; do
    i=$((i+1))
    timeout -k 10 200 $B $flags > $S/b1_${i}_$rep.json 2> $S/b1_${i}_$rep.err
    python3 -c "import json; d=json.load(open('$S/b1_${i}_$rep.json')); print('[$flags]', round(d['ms_per_step'],2), 'ms per frame', d['config']['sam_chunk_workgroup_caps'], d['config']['sam_waits_for_prefill'])" | tee -a $S/summary.txt
  done
done
