#!/bin/bash
# usage: tools/build_window_variant.sh name [-DFLAG ...]  ->  2handedafforder_amd/lib/libhaff_win_<name>.so (experiment builds of
# window_attention.hip alone; -DHAFF_TUNING turns on the ablation hooks the product build does not carry)
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
name="$1"; shift
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=fast -shared -DHAFF_TUNING "$@" \
  "$ROOT/2handedafforder_amd/csrc/window_attention.hip" -o "$ROOT/2handedafforder_amd/lib/libhaff_win_${name}.so"
