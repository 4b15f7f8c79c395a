#!/bin/bash
# Round 6: how the chained decode launch and the encoder share the chip at a few frames per step. Same box, alternating.
R=$GRAFT_REPO_ROOT
S=$R/gpurun_out/r6chain2
mkdir -p $S
cd $R
: > $S/summary.txt
run() {  # name, flags
  timeout -k 10 400 python3 bench.py $2 --no-parity --no-cpu-baseline --no-b1 > $S/$1.json 2> $S/$1.err
  python3 -c "import json; d=json.load(open('$S/$1.json')); print('$1', '[$2]', round(d['ms_per_step'],2), 'ms per step,', round(d['value'],2), d['unit'], d['config'].get('sam_chunk_workgroup_caps'), d['config'].get('sam_waits_for_prefill'))" | tee -a $S/summary.txt
}
for c in "b2 --batch 2 --steps 20 --warmup 4" "b4 --batch 4 --steps 20 --warmup 4" "b8 --batch 8 --steps 16 --warmup 4" "13b_b8 --config 13b --batch 8 --sam-chunk 8 --steps 10 --warmup 3"; do
  set -- $c; n=$1; shift
  run ${n}_five "$* --no-decode-chain"
  run ${n}_chain_plan "$*"
  run ${n}_chain_nocaps "$* --sam-caps off"
  run ${n}_chain_nocaps_wait "$* --sam-caps off --sam-waits-for-prefill on"
  run ${n}_chain_encfirst "$* --sam-beside-decode off --sam-caps off"
  run ${n}_five_encfirst "$* --sam-beside-decode off --sam-caps off --no-decode-chain"
done
