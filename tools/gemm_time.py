#!/usr/bin/env python3
"""HIP-event time of the forward projection kernel alone (prepared weight copies, 30 launches after 10), both arithmetics, at one
shape -- for same-box A/B runs of variant libraries (NPI_GNN_LIB).  usage: tools/gemm_time.py [rows [K [N]]]"""
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
res = []
for name, fn in (("bf16x3", lambda: NF.linear_fwd(a, w, ws=ws3, out=out)), ("fp16x2", lambda: NF.linear_fwd(a, w, ws=ws2, a_scales=sc, out=out))):
    for _ in range(10):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30):
        fn()
    e1.record()
    torch.cuda.synchronize()
    res.append(f"{name} {e0.elapsed_time(e1) / 30:.4f} ms")
ref = (a[:2048].double() @ w.double()).float()
err = float((out[:2048] - ref).abs().max() / ref.abs().max())
print(os.path.basename(os.environ.get("NPI_GNN_LIB", "default")), M, K, N, "|", "  ".join(res), f"| fp16x2 max err / max {err:.2e}")
