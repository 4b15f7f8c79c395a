#!/bin/bash
# Round 6: the chained decode launch (csrc/decode_chain.hip) end to end, same box, alternating: one frame per step (BASELINE configs[1]),
# 7B at 4 / 8 frames, 13B at 8 frames (configs[4]).    /usr/local/graft/bin/gpurun --timeout 1200 -- 'bash tools/r6_chain_ab.sh'
R=$GRAFT_REPO_ROOT
S=$R/gpurun_out/r6chain
mkdir -p $S
cd $R
: > $S/summary.txt
run() {  # name, flags
  timeout -k 10 400 python3 bench.py $2 --no-parity --no-cpu-baseline --no-b1 > $S/$1.json 2> $S/$1.err
  python3 -c "import json; d=json.load(open('$S/$1.json')); print('$1', '[$2]', round(d['ms_per_step'],2), 'ms per step,', round(d['value'],2), d['unit'])" | tee -a $S/summary.txt
}
for rep in 1 2; do
  for c in "b1 --batch 1 --steps 30 --warmup 5" "b4 --batch 4 --steps 20 --warmup 4" "b8 --batch 8 --steps 16 --warmup 4" "13b_b8 --config 13b --batch 8 --sam-chunk 8 --steps 10 --warmup 3"; do
    set -- $c; n=$1; shift
    run ${n}_chain_$rep "$*"
    run ${n}_five_$rep "$* --no-decode-chain"
  done
done
