#!/bin/bash
# Round 6, item 2: the fused fp32 residual streams — tests, the parity Pareto (Gaussian and two-plateau fields at full size) and
# the same-box throughput A/B.   gpurun --timeout 1200 -- 'bash tools/r6_stream_ab.sh'
set -e
R=$GRAFT_REPO_ROOT
S=$R/gpurun_out/r6s
mkdir -p $S
cd $R
python3 tools/overlap_probe.py > $S/overlap_probe.txt 2>&1
echo "[1] tests"
python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "rowstats" > $S/tests.txt 2>&1
python -m pytest tests/test_lisa_gpu.py -x -q -s -m gpu -k "vith_width or fp32_residual" >> $S/tests.txt 2>&1
tail -3 $S/tests.txt
echo "[2] throughput A/B, same box"
B="python3 bench.py --steps 8 --warmup 3 --no-parity --no-cpu-baseline --no-b1"
$B > $S/ab_bf16.json 2> $S/ab_bf16.err
$B --fp32-stream both > $S/ab_fused_both.json 2> $S/ab_fused_both.err
$B --fp32-stream both --neck-f32 > $S/ab_fused_both_neck.json 2> $S/ab_fused_both_neck.err
$B --fp32-stream sam > $S/ab_fused_sam.json 2> $S/ab_fused_sam.err
$B --fp32-stream both --unfused-fp32-stream > $S/ab_unfused_both.json 2> $S/ab_unfused_both.err
$B > $S/ab_bf16_again.json 2> $S/ab_bf16_again.err
for f in $S/ab_*.json; do python3 -c "import json,sys; d=json.load(open('$f')); print('$f'.split('/')[-1], round(d['value'],2), d['roofline']['frac'])"; done
echo "[3] parity Pareto at full size"
python3 tools/full_frame_parity.py --out $S/pareto_gaussian.json > /dev/null 2> $S/pareto_gaussian.err
python3 tools/full_frame_parity.py --field two_plateau --out $S/pareto_two_plateau.json > /dev/null 2> $S/pareto_two_plateau.err
python3 - <<'PY'
import json, os
S = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/r6s"
for f in ("pareto_gaussian", "pareto_two_plateau"):
    p = json.load(open(f"{S}/{f}.json"))["parity"]
    for k in ("fp32", "bf16", "bf16_fp32_stream", "bf16_fp32_stream_f32neck"):
        if k in p:
            print(f, k, "IoU L/R %.5f %.5f" % (p[k]["left"]["mask_iou"], p[k]["right"]["mask_iou"]), "rel err %.2e" % p[k]["logit_max_rel_err"],
                  "emb rms %.2e" % p[k]["stage_rel_err"]["image_embedding_rms"])
PY
echo done
