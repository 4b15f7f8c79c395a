set -e
mkdir -p gpurun_out/r5
B="python bench.py --steps 8 --warmup 3 --no-parity --no-cpu-baseline --no-b1"
export HAFF_LIB_PATH=$PWD/2handedafforder_amd/lib/libhaff_hip_tuning.so
HAFF_GEMM_NO_SPEC=1 $B > gpurun_out/r5/ab_nospec.json 2>/dev/null
HAFF_GEMM_NO_SPEC=0 $B > gpurun_out/r5/ab_spec.json 2>/dev/null
HAFF_GEMM_NO_SPEC=1 $B > gpurun_out/r5/ab_nospec2.json 2>/dev/null
HAFF_GEMM_NO_SPEC=0 $B > gpurun_out/r5/ab_spec2.json 2>/dev/null
unset HAFF_LIB_PATH
python bench.py --steps 10 --warmup 3 --no-parity --no-cpu-baseline --config 13b --batch 8 --sam-chunk 8 > gpurun_out/r5/bench13b_b8.json 2>/dev/null
python bench.py --mode train --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/r5/train7b_b8.json 2>/dev/null
