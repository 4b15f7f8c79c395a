#!/bin/bash
# One gpurun call: the decode step (tools/decode_step_bench.py) under every libhaff_chain_<name>.so variant present, alternating with the
# product library; results in gpurun_out/chain/ab.txt
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/chain
O=$R/gpurun_out/chain/ab.txt
: > $O
B=${BATCHES:-1,8}
for rep in 1 2; do
  echo "== product (rep $rep)" >> $O; python3 $R/tools/decode_step_bench.py --batches $B 2>&1 | grep batch >> $O
  echo "== product, five launches (rep $rep)" >> $O; python3 $R/tools/decode_step_bench.py --batches $B --no-chain 2>&1 | grep batch >> $O
  for f in $R/2handedafforder_amd/lib/libhaff_chain_*.so; do
    n=$(basename $f .so); echo "== $n (rep $rep)" >> $O
    HAFF_LIB_PATH=$f timeout -k 10 120 python3 $R/tools/decode_step_bench.py --batches $B 2>&1 | grep batch >> $O
  done
done
cat $O
