#!/bin/bash
# usage: tools/build_gemm_variant.sh name [-DFLAG ...]  ->  2handedafforder_amd/lib/libhaff_gemm_<name>.so (experiment builds;
# -DHAFF_TUNING turns on the ablation / trace / environment hooks the product build does not carry)
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
name="$1"; shift
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=fast -shared -DHAFF_TUNING "$@" \
  "$ROOT/2handedafforder_amd/csrc/gemm_bf16.hip" -o "$ROOT/2handedafforder_amd/lib/libhaff_gemm_${name}.so"
