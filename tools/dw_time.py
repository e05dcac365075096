#!/usr/bin/env python3
"""HIP-event time of the weight-gradient GEMM alone (dW = A^T dC over M rows, both grid regimes) at one shape, and its error against
fp64 -- for same-box A/B runs of variant libraries (NPI_GNN_LIB).  usage: tools/dw_time.py [rows [K [N]]]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from npi_gnn_amd import functional as NF
dev = torch.device("cuda:0")
M, K, N = (int(v) for v in (sys.argv[1:4] + ["1000000", "256", "256"][len(sys.argv) - 1:]))
g = torch.Generator(device=dev).manual_seed(1)
a = torch.randn(M, K, device=dev, generator=g)
dc = torch.randn(M, N, device=dev, generator=g)
res = []
acs, dcs = NF.col_scales(a), NF.col_scales(dc)
for name, kw in (("alone", {}), ("shared", dict(shared=True)), ("alone fp16x2", dict(a_cs=acs, dc_cs=dcs)), ("shared fp16x2", dict(shared=True, a_cs=acs, dc_cs=dcs))):
    for _ in range(5):
        dw, db = NF.linear_bwd_weight(a, dc, **kw)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        dw, db = NF.linear_bwd_weight(a, dc, **kw)
    e1.record()
    torch.cuda.synchronize()
    res.append(f"{name} {e0.elapsed_time(e1) / 20:.4f} ms")
ref = a[:, :8].double().t() @ dc.double()
err = float((dw[:8].double() - ref).abs().max() / ref.abs().max())
print(os.path.basename(os.environ.get("NPI_GNN_LIB", "default")), M, K, N, "|", "  ".join(res), f"| dW max err / max {err:.2e}; db err {float((db.double() - dc.double().sum(0)).abs().max()):.2e}")
