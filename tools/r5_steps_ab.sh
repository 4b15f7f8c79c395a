set -e
mkdir -p gpurun_out/r5
B="python bench.py --no-parity --no-cpu-baseline --no-b1"
$B --steps 6 --warmup 2 > gpurun_out/r5/st6_a.json 2>/dev/null
$B --steps 20 --warmup 5 > gpurun_out/r5/st20_a.json 2>/dev/null
$B --steps 6 --warmup 2 > gpurun_out/r5/st6_b.json 2>/dev/null
$B --steps 20 --warmup 5 > gpurun_out/r5/st20_b.json 2>/dev/null
$B --steps 40 --warmup 5 > gpurun_out/r5/st40_a.json 2>/dev/null
