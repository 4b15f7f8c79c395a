set -e
mkdir -p gpurun_out/r5
B="python bench.py --steps 6 --warmup 2 --no-parity --no-cpu-baseline --no-b1"
$B > gpurun_out/r5/ab_hm.json 2>/dev/null
$B --token-major-windows > gpurun_out/r5/ab_tm.json 2>/dev/null
$B > gpurun_out/r5/ab_hm2.json 2>/dev/null
$B --token-major-windows > gpurun_out/r5/ab_tm2.json 2>/dev/null
for cfg in 7b 13b; do for b in 8 16; do for m in on off; do
  python bench.py --steps 10 --warmup 3 --no-parity --no-cpu-baseline --no-b1 --config $cfg --batch $b --sam-chunk $b --sam-beside-decode $m > gpurun_out/r5/sched_${cfg}_b${b}_${m}.json 2>/dev/null
done; done; done
