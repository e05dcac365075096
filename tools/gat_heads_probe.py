#!/usr/bin/env python3
"""GATConv with H heads at the C4 shape (H x C = 256): ms per layer step; under rocprofv3 --kernel-trace its kernels.
usage: tools/gat_heads_probe.py [heads] [steps]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import npi_gnn_amd as npi
from npi_gnn_amd.synth import bipartite_edge_index
dev = torch.device("cuda:0")
H = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n = int(sys.argv[2]) if len(sys.argv) > 2 else 5
N, E, F = 1_000_000, 20_000_000, 256
g = npi.CSRGraph(bipartite_edge_index(N, E).to(dev), N); _ = g.by_src
conv = npi.GATConv(F, F // H, heads=H).to(dev)
x = torch.randn(N, F, device=dev).requires_grad_(True)
go = torch.randn(N, F, device=dev)
def step():
    for p in conv.parameters(): p.grad = None
    x.grad = None
    conv(x, g).backward(go)
for _ in range(3): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(n): step()
torch.cuda.synchronize()
print(f"GATConv {H} heads: {(time.perf_counter() - t0) / n * 1e3:.3f} ms per step")
