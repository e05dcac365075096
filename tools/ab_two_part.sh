#!/bin/bash
# A/B of the two-part-table select in the segsum gather (run on the GPU box from the repo root)
for i in 1 2; do
  echo "== default (two-part select compiled in)"; python3 tools/kernel_bench.py --seg --rounds 12 | head -2
  echo "== one_table variant"; NPI_GNN_LIB=$(pwd)/npi_gnn_amd/build/variants/lib_one_table.so python3 tools/kernel_bench.py --seg --rounds 12 | head -2
done
