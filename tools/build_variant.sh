#!/bin/bash
# A variant library for same-box A/B measurements: ONE translation unit recompiled with extra flags, linked with the shipped
# objects of the others, loaded through NPI_GNN_LIB.   usage: tools/build_variant.sh <name> <source.hip> [hipcc flags...]
set -eu
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
name="$1"; src="$2"; shift 2
B="$ROOT/npi_gnn_amd/build"
mkdir -p "$B/variants"
obj="$B/variants/${src%.hip}_$name.o"
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -fno-gpu-rdc -Wall -Wno-unused-function "$@" -c "$ROOT/npi_gnn_amd/csrc/$src" -o "$obj"
objs=""
for o in "$B"/*.o; do
    [ "$(basename "$o")" = "${src%.hip}.o" ] && o="$obj"
    objs="$objs $o"
done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o "$B/variants/libnpi_gnn_$name.so" $objs
echo "$B/variants/libnpi_gnn_$name.so"
