#!/bin/bash
# A variant library for same-box A/B measurements: ONE translation unit recompiled with extra flags, linked with the shipped
# objects of the others, loaded through NPI_GNN_LIB.   usage: tools/build_variant.sh <name> <source.hip> [hipcc flags...]
# The measurement switches of the projection GEMMs (cycle stamps, no-store / no-split timing builds: -DNPI_WS_PROBE=<bits>,
# -DNPI_DW_PROBE=<bits>) are NOT in the product source: apply tools/micro/gemm_f32_probes.patch to a copy first --
#   PATCH=tools/micro/gemm_f32_probes.patch tools/build_variant.sh stamps gemm_f32.hip -DNPI_WS_PROBE=16
set -eu
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
name="$1"; src="$2"; shift 2
B="$ROOT/npi_gnn_amd/build"
mkdir -p "$B/variants"
obj="$B/variants/${src%.hip}_$name.o"
in="$ROOT/npi_gnn_amd/csrc/$src"
if [ -n "${PATCH:-}" ]; then      # a patched COPY beside the original (its includes resolve), never the product file
    in="$ROOT/npi_gnn_amd/csrc/.variant_$name.hip"
    cp "$ROOT/npi_gnn_amd/csrc/$src" "$in"
    patch -s "$in" < "$ROOT/$PATCH"
    trap 'rm -f "$in"' EXIT
fi
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -fno-gpu-rdc -Wall -Wno-unused-function "$@" -c "$in" -o "$obj"
objs=""
for o in "$B"/*.o; do
    [ "$(basename "$o")" = "${src%.hip}.o" ] && o="$obj"
    objs="$objs $o"
done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o "$B/variants/libnpi_gnn_$name.so" $objs
echo "$B/variants/libnpi_gnn_$name.so"
