#!/bin/bash
# Round-5 evidence set, one gpurun call: rocprofv3 kernel stats of the inference step (one stream), of the fine-tune step and of the
# 13B line, the two-pass HBM-traffic counters of the GEMM family, MFMA / wait counters of the specialised GEMM tile and of the window
# attention on head-major planes, the batch-1 phase times, the decode-step table and the final bench lines (the default line carries
# the full-depth parity frame and the CPU baseline). Summaries under gpurun_out/r5e (copied into profiles/ afterwards); raw traces
# stay on the box.       /usr/local/graft/bin/gpurun --timeout 1200 -- 'bash tools/r5_evidence.sh'
set -e
R=$GRAFT_REPO_ROOT
S=$R/gpurun_out/r5e
O=/tmp/r5e_raw
mkdir -p $O $S
cd /tmp && export TMPDIR=/tmp
echo "[1] kernel stats, inference step, one stream"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_infer -o x -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-parity --no-b1 --single-stream > $S/infer_under_rocprof.txt 2>&1
cp $(find $O/prof_infer -name "*kernel_stats.csv" | head -1) $S/infer_single_stream_kernel_stats.csv
echo "[2] kernel stats, fine-tune step, one stream"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_train -o x -- python3 $R/bench.py --mode train --steps 5 --warmup 2 --single-stream --no-cpu-baseline > $S/train_under_rocprof.txt 2>&1
cp $(find $O/prof_train -name "*kernel_stats.csv" | head -1) $S/train_single_stream_kernel_stats.csv
echo "[3] kernel stats, 13B at 8 frames, one stream"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_13b -o x -- python3 $R/bench.py --config 13b --batch 8 --sam-chunk 8 --steps 4 --warmup 2 --no-cpu-baseline --no-parity --no-b1 --single-stream > $S/bench13b_under_rocprof.txt 2>&1
cp $(find $O/prof_13b -name "*kernel_stats.csv" | head -1) $S/bench13b_b8_single_stream_kernel_stats.csv
echo "[4] FETCH_SIZE / WRITE_SIZE passes"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o x -- python3 $R/bench.py --batch 64 --steps 1 --warmup 1 --no-cpu-baseline --no-b1 --no-parity > $S/pmc_fetch.txt 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o x -- python3 $R/bench.py --batch 64 --steps 1 --warmup 1 --no-cpu-baseline --no-b1 --no-parity > $S/pmc_write.txt 2>&1
python3 $R/tools/pmc_traffic.py $(find $O/pmc_fetch -name '*counter_collection.csv' | head -1) $(find $O/pmc_write -name '*counter_collection.csv' | head -1) 2HandedAfforder-7B 64 $S/pmc_gemm_traffic.json
echo "[5] window attention counters (head-major planes come from the encoder; the stand-alone driver uses the token-major views)"
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 --kernel-trace --output-format csv -d $O/pmc_win_mfma -o x -- python3 $R/tools/window_one.py 32 3 > $S/pmc_win_mfma.txt 2>&1
cp $(find $O/pmc_win_mfma -name "*counter_collection.csv" | head -1) $S/pmc_window_attn_mfma.csv
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $O/pmc_win_lds -o x -- python3 $R/tools/window_one.py 32 3 > $S/pmc_win_lds.txt 2>&1
cp $(find $O/pmc_win_lds -name "*counter_collection.csv" | head -1) $S/pmc_window_attn_valu_lds.csv
echo "[6] GEMM MFMA-busy, 131072x3840x1280 (specialised bias instance) and 131072x5120x1280 GELU"
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 --kernel-trace --output-format csv -d $O/pmc_gemm_mfma -o x -- python3 $R/tools/gemm_one.py 131072 3840 1280 > $S/pmc_gemm_mfma.txt 2>&1
cp $(find $O/pmc_gemm_mfma -name "*counter_collection.csv" | head -1) $S/pmc_gemm256_131072x3840x1280_mfma.csv
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 --kernel-trace --output-format csv -d $O/pmc_gemm_mfma2 -o x -- python3 $R/tools/gemm_one.py 131072 5120 1280 0 5 gelu > $S/pmc_gemm_mfma2.txt 2>&1
cp $(find $O/pmc_gemm_mfma2 -name "*counter_collection.csv" | head -1) $S/pmc_gemm256_131072x5120x1280_gelu_mfma.csv
cd $R
echo "[7] batch-1 phases, decode steps, attention benches"
python3 tools/b1_events.py > $S/b1_events.txt 2>&1
python3 tools/decode_step_bench.py > $S/decode_step_bench.txt 2>&1
python3 tools/attn_bench.py > $S/attn_bench.txt 2>&1
FRAMES=32 python3 tools/window_attn_bench.py > $S/window_attn_bench.txt 2>&1
echo "[8] bench lines"
python3 bench.py --config 13b --batch 8 --sam-chunk 8 --no-cpu-full-frame --no-parity > $S/bench13b.json 2> $S/bench13b.err
python3 bench.py --mode train --steps 5 --warmup 2 > $S/bench_train.json 2> $S/bench_train.err
python3 bench.py --steps 10 --warmup 3 > $S/bench_default.json 2> $S/bench_default.err
echo done
