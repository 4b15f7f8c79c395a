#!/bin/bash
# same-box A/B of two library builds (HAFF_LIB_PATH): tools/build/libhaff_head.so vs the tree's
mkdir -p gpurun_out/r5c
for rep in 1 2 3; do
  for lib in head new; do
    if [ $lib = head ]; then export HAFF_LIB_PATH=$GRAFT_REPO_ROOT/tools/build/libhaff_head.so; else unset HAFF_LIB_PATH; fi
    for cfg in "$@"; do
      python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-parity --no-b1 $cfg > gpurun_out/r5c/ab4.json 2> gpurun_out/r5c/ab4.err || exit 1
      python3 -c "
import json
d=json.load(open('gpurun_out/r5c/ab4.json')); print('$lib [$cfg]', round(d['value'],2), round(d['ms_per_step'],1))"
    done
  done
done
