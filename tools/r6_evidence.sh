#!/bin/bash
# Round-6 evidence set, one gpurun call: the pruning test, rocprofv3 kernel stats of the inference step (one stream), the two-pass
# HBM-traffic counters of the GEMM family (stamped with the csrc hash), batch-1 phases, the 13B / fine-tune (7B and 13B) lines and the
# driver-style default line (compact stdout line + bench_detail.json). Summaries under gpurun_out/r6e (copied into profiles/).
#       /usr/local/graft/bin/gpurun --timeout 1200 -- 'bash tools/r6_evidence.sh'
set -e
R=$GRAFT_REPO_ROOT
S=$R/gpurun_out/r6e
O=/tmp/r6e_raw
mkdir -p $O $S
cd $R
echo "[0] tests touched since the full run"
python -m pytest tests/test_lisa_gpu.py -x -q -m gpu -k "pruning or ragged or stream_schedules" > $S/tests.txt 2>&1
tail -2 $S/tests.txt
cd /tmp && export TMPDIR=/tmp
echo "[1] kernel stats, inference step, one stream"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_infer -o x -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-parity --no-b1 --single-stream > $S/infer_under_rocprof.json 2> $S/infer_under_rocprof.err
cp $(find $O/prof_infer -name "*kernel_stats.csv" | head -1) $S/infer_single_stream_kernel_stats.csv
echo "[2] FETCH_SIZE / WRITE_SIZE passes"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o x -- python3 $R/bench.py --batch 64 --steps 1 --warmup 1 --no-cpu-baseline --no-b1 --no-parity > $S/pmc_fetch.txt 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o x -- python3 $R/bench.py --batch 64 --steps 1 --warmup 1 --no-cpu-baseline --no-b1 --no-parity > $S/pmc_write.txt 2>&1
python3 $R/tools/pmc_traffic.py $(find $O/pmc_fetch -name '*counter_collection.csv' | head -1) $(find $O/pmc_write -name '*counter_collection.csv' | head -1) 2HandedAfforder-7B 64 $S/pmc_gemm_traffic.json
cp $S/pmc_gemm_traffic.json $R/profiles/pmc_gemm_traffic.json      # the default line below reads it (stamp == this tree)
cd $R
echo "[3] batch-1 phases"
python3 tools/b1_events.py > $S/b1_events.txt 2>&1
echo "[4] bench lines"
python3 bench.py --config 13b --batch 8 --sam-chunk 8 --no-cpu-full-frame --no-parity > $S/bench13b.json 2> $S/bench13b.err; cp gpurun_out/bench_detail.json $S/bench13b_detail.json
python3 bench.py --mode train --steps 5 --warmup 2 > $S/bench_train7b.json 2> $S/bench_train7b.err; cp gpurun_out/bench_detail.json $S/bench_train7b_detail.json
python3 bench.py --mode train --config 13b --steps 5 --warmup 2 --no-cpu-baseline > $S/bench_train13b.json 2> $S/bench_train13b.err; cp gpurun_out/bench_detail.json $S/bench_train13b_detail.json
S0=$SECONDS
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $S/bench_default.json 2> $S/bench_default.err; cp gpurun_out/bench_detail.json $S/bench_default_detail.json
echo "default line: wall $((SECONDS-S0)) s, $(wc -c < $S/bench_default.json) bytes" | tee $S/bench_default.wall
for f in bench13b bench_train7b bench_train13b bench_default; do python3 -c "import json; d=json.load(open('$S/$f.json')); print('$f', round(d['value'],2), d['unit'], round(d['ms_per_step'],1), 'ms frac', d['roofline']['frac'], 'traffic', d['roofline'].get('traffic'))"; done
echo done
