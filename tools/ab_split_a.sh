#!/bin/bash
# A/B of the A-operand LDS staging in gemm_split_ws_kernel (16-byte conflict-free stores vs 8-byte stores); GPU box
for i in 1 2; do
  echo "== 8-byte stores (default)"; timeout -k 5 200 python3 tools/kernel_bench.py --gemm --rounds 10 2>/dev/null | head -2
  echo "== 16-byte stores (tools/build_variant.sh split_a16 -DNPI_SPLIT_A16=1)"; NPI_GNN_LIB=$(pwd)/npi_gnn_amd/build/variants/lib_split_a16.so timeout -k 5 200 python3 tools/kernel_bench.py --gemm --rounds 10 2>/dev/null | head -2
done
