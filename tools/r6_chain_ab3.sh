R=$GRAFT_REPO_ROOT
S=$R/gpurun_out/r6chain3
mkdir -p $S
cd $R
run() {
  timeout -k 10 400 python3 bench.py $2 --no-parity --no-cpu-baseline --no-b1 > $S/$1.json 2> $S/$1.err
  python3 -c "import json; d=json.load(open('$S/$1.json')); print('$1', '[$2]', round(d['ms_per_step'],2), 'ms per step', d['config'].get('sam_chunk_workgroup_caps'), d['config'].get('sam_waits_for_prefill'))"
}
for rep in 1 2; do
for c in "b1 --batch 1 --steps 30 --warmup 5" "b3 --batch 3 --steps 20 --warmup 4"; do
  set -- $c; n=$1; shift
  run ${n}_five "$* --no-decode-chain"
  run ${n}_chain_late "$*"
  run ${n}_chain_encfirst "$* --sam-beside-decode off --sam-caps off"
  run ${n}_chain_single "$* --single-stream"
done
done
