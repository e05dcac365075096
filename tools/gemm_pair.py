#!/usr/bin/env python3
"""The two forward projection kernels (six bf16 products / three fp16 products per tile pair) at ONE shape, 20 launches each: the
program behind the MFMA-utilisation PMC pass (tools/mfma_util_summary.py).  usage: tools/gemm_pair.py [rows [K [N]]]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from npi_gnn_amd import functional as NF
dev = torch.device("cuda:0")
M, K, N = (int(v) for v in (sys.argv[1:4] + ["1000000", "256", "256"][len(sys.argv) - 1:]))
g = torch.Generator(device=dev).manual_seed(1)
a = torch.randn(M, K, device=dev, generator=g)
w = torch.randn(K, N, device=dev, generator=g) / K ** 0.5
out = torch.empty(M, N, device=dev)
sc = NF.row_scales(a)
ws3, _ = NF.prepare_weight(w, backward=False)
ws2, _ = NF.prepare_weight(w, backward=False, f16=True)
for _ in range(20):
    NF.linear_fwd(a, w, ws=ws3, out=out)
for _ in range(20):
    NF.linear_fwd(a, w, ws=ws2, a_scales=sc, out=out)
torch.cuda.synchronize()
print("done", M, K, N)
