#!/bin/bash
# usage: tools/build_chain_variant.sh name [-DCH_... ...]  ->  2handedafforder_amd/lib/libhaff_chain_<name>.so: the product library with
# csrc/decode_chain.hip rebuilt under the given knobs (select it with HAFF_LIB_PATH=...). Needs the product build's objects (make).
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
C="$ROOT/2handedafforder_amd/csrc"
name="$1"; shift
mkdir -p /tmp/chainvar
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=fast -Wno-unused-result "$@" -c "$C/decode_chain.hip" -o /tmp/chainvar/decode_chain_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/2handedafforder_amd/lib/libhaff_chain_${name}.so" \
  $(ls $C/build/*.o | grep -v decode_chain.o) /tmp/chainvar/decode_chain_$name.o
echo "built libhaff_chain_${name}.so"
