#!/bin/bash
# where does gemm_dw_split_kernel spend its time?  dW alone at C4 (1M x 256 x 256) with the default library and the probe
# variants of tools/build_variant.sh (-DNPI_DW_PROBE=1: no start stagger, 2: cheap split, 4: one MFMA per product tile)
for v in default dw_nostagger dw_cheapsplit dw_onemfma; do
  if [ $v = default ]; then unset NPI_GNN_LIB; else export NPI_GNN_LIB=$(pwd)/npi_gnn_amd/build/variants/lib_$v.so; fi
  echo "== $v: $(timeout -k 5 200 python3 tools/kernel_bench.py --gemm --rounds 8 2>/dev/null | grep bwd_weight | head -1)"
done
