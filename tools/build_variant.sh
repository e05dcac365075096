#!/bin/bash
# build an experimental variant of the library: tools/build_variant.sh <name> [-DFLAG=...]
set -e
name=$1; shift
d=npi_gnn_amd/build/variants; mkdir -p $d
objs=""
for f in csr_build segsum gemm_f32 graph_ops gat pool subgraph head; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -fno-gpu-rdc "$@" -c npi_gnn_amd/csrc/$f.hip -o $d/${name}_$f.o &
  objs="$objs $d/${name}_$f.o"
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $d/lib_$name.so $objs
echo $d/lib_$name.so
