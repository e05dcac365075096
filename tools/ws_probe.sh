#!/bin/bash
# where the split-bf16 GEMM's time goes: timing-only variants (tools/build_variant.sh wsp<bits> -DNPI_WS_PROBE=<bits>)
echo "== default"; timeout -k 5 200 python3 tools/kernel_bench.py --gemm --rounds 10 2>/dev/null | head -2
for v in "$@"; do
  echo "== probe $v"; NPI_GNN_LIB=$(pwd)/npi_gnn_amd/build/variants/lib_wsp$v.so timeout -k 5 200 python3 tools/kernel_bench.py --gemm --rounds 10 2>/dev/null | head -2
done
