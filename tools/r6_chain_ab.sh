#!/bin/bash
# Round 6: the chained decode launch (csrc/decode_chain.hip) — evidence set, one gpurun call, same box, alternating:
#   [1] decode step alone (hipGraph replay): chained vs five launches per layer, 1 / 2 / 4 / 8 rows (7B), 8 rows (13B)
#   [2] end to end: one / two / three frames per step (BASELINE configs[1] and neighbours): chained (the default there) vs five launches;
#       4 / 8 frames and 13B at 8 frames (configs[4]): the default (five launches beside the capped encoder) vs the chain forced on
#   [3] per-kernel durations of the five-launch step (rocprofv3) and the per-stage timeline + placement of the chained one (-DCH_TRACE / -DCH_PLACE builds)
#   [4] repeatability of evaluate() in every combination of {two streams, one} x {hipGraph, eager} x {chain, stage by stage, five launches}
#       /usr/local/graft/bin/gpurun --timeout 1200 -- 'bash tools/r6_chain_ab.sh'
R=$GRAFT_REPO_ROOT
S=$R/gpurun_out/r6chain
mkdir -p $S
cd $R
echo "[1] decode step alone" | tee $S/decode_step.txt
for rep in 1 2; do
  echo "chained launch (rep $rep)" >> $S/decode_step.txt; python3 tools/decode_step_bench.py --batches 1,2,4,8 2>&1 | grep batch >> $S/decode_step.txt
  echo "five launches per layer (rep $rep)" >> $S/decode_step.txt; python3 tools/decode_step_bench.py --batches 1,2,4,8 --no-chain 2>&1 | grep batch >> $S/decode_step.txt
done
echo "13B chained" >> $S/decode_step.txt; python3 tools/decode_step_bench.py --config 13b --batches 1,8 2>&1 | grep batch >> $S/decode_step.txt
echo "13B five launches per layer" >> $S/decode_step.txt; python3 tools/decode_step_bench.py --config 13b --batches 1,8 --no-chain 2>&1 | grep batch >> $S/decode_step.txt
cat $S/decode_step.txt
echo "[2] end to end" | tee $S/end_to_end.txt
run() {  # name, flags
  timeout -k 10 400 python3 bench.py $2 --no-parity --no-cpu-baseline --no-b1 > $S/$1.json 2> $S/$1.err
  python3 -c "import json; d=json.load(open('$S/$1.json')); print('$1', '[$2]', round(d['ms_per_step'],2), 'ms per step,', round(d['value'],2), d['unit'], 'caps', d['config'].get('sam_chunk_workgroup_caps'), 'decode_chain', d['config'].get('decode_chain'))" | tee -a $S/end_to_end.txt
}
for rep in 1 2; do
  for c in "b1 --batch 1 --steps 30 --warmup 5" "b2 --batch 2 --steps 20 --warmup 4" "b3 --batch 3 --steps 20 --warmup 4"; do
    set -- $c; n=$1; shift
    run ${n}_chain_$rep "$*"
    run ${n}_five_$rep "$* --no-decode-chain"
    run ${n}_five_encfirst_$rep "$* --no-decode-chain --sam-beside-decode off --sam-caps off"
    run ${n}_chain_late_$rep "$* --sam-beside-decode on"
  done
done
for c in "b4 --batch 4 --steps 20 --warmup 4" "b8 --batch 8 --steps 16 --warmup 4" "13b_b8 --config 13b --batch 8 --sam-chunk 8 --steps 10 --warmup 3"; do
  set -- $c; n=$1; shift
  run ${n}_default "$*"
  run ${n}_chain_forced "$* --decode-chain on"
done
echo "[3] traces"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/dect5 -o x -- python3 $R/tools/decode_step_bench.py --batches 1 --no-chain > /dev/null 2>&1
python3 $R/tools/decode_trace.py $(find /tmp/dect5 -name "*kernel_trace.csv" | head -1) 1650 > $S/five_launch_kernels_b1.txt 2>&1
cd $R
bash tools/build_chain_variant.sh trace -DCH_TRACE -DCH_PLACE > /dev/null 2>&1
HAFF_LIB_PATH=$R/2handedafforder_amd/lib/libhaff_chain_trace.so python3 tools/chain_trace.py 1 8 > $S/chain_timeline_b1.txt 2>&1
HAFF_LIB_PATH=$R/2handedafforder_amd/lib/libhaff_chain_trace.so python3 tools/chain_trace.py 8 8 > $S/chain_timeline_b8.txt 2>&1
rm -f $R/2handedafforder_amd/lib/libhaff_chain_trace.so
tail -8 $S/chain_timeline_b1.txt
echo "[4] repeatability"
python3 tools/chain_stress.py 1 6 > $S/chain_stress_b1.txt 2>&1
python3 tools/chain_stress.py 3 6 > $S/chain_stress_b3.txt 2>&1
grep -c "0 of 5" $S/chain_stress_b1.txt $S/chain_stress_b3.txt
echo done
