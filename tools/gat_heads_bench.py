#!/usr/bin/env python3
"""GATConv fwd+bwd at the C4 shape for several head counts (H x C = 256): ms per layer, and the kernels under rocprofv3."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import npi_gnn_amd as npi
from npi_gnn_amd.synth import bipartite_edge_index

dev = torch.device("cuda:0")
N, E, F = 1_000_000, 20_000_000, 256
heads = [int(h) for h in (sys.argv[1:] or ["1", "2", "4", "8"])]
ei = bipartite_edge_index(N, E, seed=20260310).to(dev)
g = npi.CSRGraph(ei, N); _ = g.by_src; del ei
x = torch.randn(N, F, device=dev).requires_grad_(True)
go = torch.randn(N, F, device=dev)
for H in heads:
    conv = npi.GATConv(F, F // H, heads=H).to(dev)
    def step():
        for p in conv.parameters(): p.grad = None
        x.grad = None
        conv(x, g).backward(go)
    for _ in range(3): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): step()
    torch.cuda.synchronize()
    print(f"heads {H} x {F // H}: {(time.perf_counter() - t0) / 10 * 1e3:.3f} ms per layer (fwd+bwd)")
