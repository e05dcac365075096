#!/usr/bin/env python3
"""Randomised campaign for the sharded layers on ONE GPU: W virtual ranks (the lock-step harness of tests/test_dist_gpu.py:
collectives resolved in-process) on random bipartite / arbitrary graphs, SAGE / GCN / GAT with 1-8 heads, against the
single-GPU layers of this package.  usage: tools/fuzz_dist.py [cases] [seed]"""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import npi_gnn_amd as npi
from npi_gnn_amd import dist as ND
from npi_gnn_amd.synth import bipartite_edge_index, protein_mask
import test_dist_gpu as T

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
dev = torch.device("cuda:0")
worst = 0.0
for it in range(cases):
    W = int(rng.integers(1, 9))
    kind = ["sage", "gcn", "gat1", "gat2", "gat4", "gat8"][int(rng.integers(0, 6))]
    F = 256 if kind in ("gat8",) else int(rng.choice([128, 256]))
    bip = bool(rng.random() < 0.75)
    N = int(rng.integers(200, 9000))
    E = 2 * int(rng.integers(100, 40000))
    g = torch.Generator().manual_seed(int(rng.integers(0, 2 ** 31)))
    if bip:
        E = min(E, 2 * ((N - max(1, N // 10)) * max(1, N // 10) // 3))
        ei = bipartite_edge_index(N, max(E, 2), seed=int(rng.integers(0, 2 ** 31)))
        hub = protein_mask(N)
    else:
        ei = torch.randint(0, N, (2, E // 2), generator=g)
        ei = torch.cat([ei, ei.flip(0)], dim=1)
        hub = None
    x, go = torch.randn(N, F, generator=g), torch.randn(N, F, generator=g)
    Wm, b = torch.randn(F, F, generator=g) / F ** 0.5, torch.randn(F, generator=g) * 0.1
    outs, dxs, dws = T._run_virtual(ND, W, kind, ei, N, F, x, go, Wm, b, hub, dev)
    part = ND.HubPartition(N, W, hub)
    out, dx = part.unshard(outs), part.unshard(dxs)
    if kind == "sage":
        conv = npi.SAGEConv(F, F)
    elif kind == "gcn":
        conv = npi.GCNConv(F, F)
    else:
        H = int(kind[3:])
        conv = npi.GATConv(F, F // H, heads=H)
    conv = conv.to(dev)
    with torch.no_grad():
        conv.weight.copy_(Wm); conv.bias.copy_(b)
        if kind.startswith("gat"):
            conv.att.copy_(T._att(F, int(kind[3:])))
    xr = x.to(dev).requires_grad_(True)
    ref = conv(xr, npi.CSRGraph(ei.to(dev), N))
    ref.backward(go.to(dev))
    def rel(a, r):
        return float((a - r.cpu()).abs().max() / r.abs().max().clamp(min=1e-6))
    errs = {"out": rel(out, ref.detach()), "dx": rel(dx, xr.grad), "dW": max(rel(dw, conv.weight.grad) for dw in dws)}
    m = max(errs.values())
    worst = max(worst, m)
    if m > 1e-4:
        print(f"MISMATCH case {it}: W={W} {kind} F={F} bipartite={bip} N={N} E={ei.size(1)}: {errs}")
        sys.exit(1)
print(f"{cases} cases ok, worst relative error {worst:.2e}")
