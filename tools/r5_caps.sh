#!/bin/bash
mkdir -p gpurun_out/r5c
run() { # name, config, batch, chunk, caps, wait
  python3 bench.py --config $2 --batch $3 --sam-chunk $4 --steps 6 --warmup 2 --sam-caps $5 --sam-waits-for-prefill $6 --no-cpu-baseline --no-parity --no-b1 > gpurun_out/r5c/capsv_$1.json 2> gpurun_out/r5c/capsv_$1.err || exit 1
  python3 -c "
import json
d=json.load(open('gpurun_out/r5c/capsv_$1.json')); print('$1', '$2', 'B$3', 'chunk$4', '$5', '$6', round(d['value'],2), round(d['ms_per_step'],1))"
}
run a 7b 16 16 off off
run b 7b 16 16 off on
run c 7b 16 16 224 on
run d 7b 16 16 216 on
run e 7b 16 16 192 on
run f 7b 16 16 160 on
run g 7b 16 8 off off
run h 7b 16 8 192 on
run i 7b 16 8 160 on
run j 7b 16 8 224 on
run k 7b 4 4 off off
run l 7b 4 4 off on
run m 7b 4 4 192 on
run n 7b 4 4 160 on
run o 7b 4 4 128 on
run p 7b 32 16 off off
run q 7b 32 16 256,224 off
run r 7b 32 16 256,192 off
run s 7b 32 8 off off
run t 7b 32 8 256,256,224,224 off
run u 7b 32 8 256,256,192,192 off
run v 7b 32 8 256,256,160,160 off
