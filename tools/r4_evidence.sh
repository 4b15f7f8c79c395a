#!/bin/bash
# Round-4 evidence set, one gpurun call: rocprofv3 kernel stats of the inference step (one stream) and of the fine-tune step, the
# two-pass HBM-traffic counters of the GEMM family, the LDS / MFMA counters of the global attention after the K-stride change,
# the batch-1 phase times and the 13B line. Summaries under gpurun_out/r4n (copied into profiles/ afterwards); the raw traces stay on the box.
#   /usr/local/graft/bin/gpurun --timeout 1200 -- 'bash tools/r4_evidence.sh'
set -e
R=$GRAFT_REPO_ROOT
S=$R/gpurun_out/r4n      # summaries (merged back by gpurun: <= 64 MiB)
O=/tmp/r4n_raw             # raw profiler output stays on the box
mkdir -p $O $S
cd /tmp && export TMPDIR=/tmp
echo "[1] kernel stats, inference step, one stream"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_infer -o x -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-parity --no-b1 --single-stream > $S/infer_under_rocprof.txt 2>&1
cp $(find $O/prof_infer -name "*kernel_stats.csv" | head -1) $S/infer_single_stream_kernel_stats.csv
echo "[2] kernel stats, fine-tune step, one stream"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_train -o x -- python3 $R/bench.py --mode train --steps 5 --warmup 2 --single-stream --no-cpu-baseline > $S/train_under_rocprof.txt 2>&1
cp $(find $O/prof_train -name "*kernel_stats.csv" | head -1) $S/train_single_stream_kernel_stats.csv
echo "[3] FETCH_SIZE / WRITE_SIZE passes"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o x -- python3 $R/bench.py --batch 64 --steps 1 --warmup 1 --no-cpu-baseline --no-b1 --no-parity > $S/pmc_fetch.txt 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o x -- python3 $R/bench.py --batch 64 --steps 1 --warmup 1 --no-cpu-baseline --no-b1 --no-parity > $S/pmc_write.txt 2>&1
python3 $R/tools/pmc_traffic.py $(find $O/pmc_fetch -name '*counter_collection.csv' | head -1) $(find $O/pmc_write -name '*counter_collection.csv' | head -1) 2HandedAfforder-7B 64 $S/pmc_gemm_traffic.json
echo "[4] global attention counters (rel-pos in the prologue, 32 frames)"
export FUSED=1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $O/pmc_attn_lds -o x -- python3 $R/tools/attn_one.py 32 3 > $S/pmc_attn_lds.txt 2>&1
cp $(find $O/pmc_attn_lds -name "*counter_collection.csv" | head -1) $S/pmc_attn_global_lds.csv
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 --kernel-trace --output-format csv -d $O/pmc_attn_mfma -o x -- python3 $R/tools/attn_one.py 32 3 > $S/pmc_attn_mfma.txt 2>&1
cp $(find $O/pmc_attn_mfma -name "*counter_collection.csv" | head -1) $S/pmc_attn_global_mfma.csv
unset FUSED
echo "[5] GEMM MFMA-busy, 131072x3840x1280"
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 --kernel-trace --output-format csv -d $O/pmc_gemm_mfma -o x -- python3 $R/tools/gemm_one.py 131072 3840 1280 > $S/pmc_gemm_mfma.txt 2>&1
cp $(find $O/pmc_gemm_mfma -name "*counter_collection.csv" | head -1) $S/pmc_gemm256_131072x3840x1280_mfma.csv
cd $R
echo "[6] batch-1 phases, attention / window benches, 13B"
python3 tools/b1_events.py > $S/b1_events.txt 2>&1
python3 tools/attn_bench.py > $S/attn_bench.txt 2>&1
python3 bench.py --config 13b --batch 8 --no-cpu-full-frame --no-parity > $S/bench13b.json 2> $S/bench13b.err
echo "[7] bench lines"
python3 bench.py --mode train --steps 5 --warmup 2 > $S/bench_train.json 2> $S/bench_train.err
python3 bench.py > $S/bench_default.json 2> $S/bench_default.err
echo done
