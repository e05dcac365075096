#!/bin/bash
# A/B of the MFMA issue order in the split-bf16 consumers (interleaved over tiles vs six dependent MFMAs per tile)
for i in 1 2; do
  echo "== chained (default)"; timeout -k 5 200 python3 tools/kernel_bench.py --gemm --rounds 10 2>/dev/null | head -3
  echo "== interleaved variant (tools/build_variant.sh mma_interleave -DNPI_MMA_INTERLEAVE=1)"; NPI_GNN_LIB=$(pwd)/npi_gnn_amd/build/variants/lib_mma_interleave.so timeout -k 5 200 python3 tools/kernel_bench.py --gemm --rounds 10 2>/dev/null | head -3
done
