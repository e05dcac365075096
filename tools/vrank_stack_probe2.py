#!/usr/bin/env python3
"""dX of the 3-layer stack, row by row: sharded (lock step) vs single GPU vs single GPU with permuted edges."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import npi_gnn_amd as npi
from npi_gnn_amd import dist as ND
from npi_gnn_amd.synth import bipartite_edge_index, protein_mask
from npi_gnn_amd.virtual import LockStep
dev = torch.device("cuda:0")
F, W, L = 256, 8, 3
N, E = int(sys.argv[1]), int(sys.argv[2])
ei = bipartite_edge_index(N, E, seed=2).to(dev)
g = torch.Generator().manual_seed(23)
ps = [((torch.randn(F, F, generator=g) / 16).to(dev), (torch.randn(1, 1, 2 * F, generator=g) * 0.3).to(dev), (torch.randn(F, generator=g) * 0.1).to(dev)) for _ in range(L)]
x = torch.randn(N, F, generator=g).to(dev)
go = torch.randn(N, F, generator=g).to(dev)
hub = protein_mask(N).to(dev)

def single(edges):
    graph = npi.CSRGraph(edges, N)
    xin = x.clone().requires_grad_(True)
    h = xin
    for Wm, a, b in ps:
        h = npi.gat_conv(h, graph, Wm, a, b, heads=1)
    h.backward(go)
    return h.detach(), xin.grad
o1, d1 = single(ei)
o2, d2 = single(ei[:, torch.randperm(E, generator=g).to(dev)].contiguous())
with LockStep(W) as ls:
    sgs = [ND.ShardedGraph(ei, N, r, W, dev, hub_mask=hub) for r in range(W)]
    def run(r):
        sg = sgs[r]
        xl = x[sg.own].clone().requires_grad_(True)
        h = xl
        for Wm, a, b in ps:
            h = ND.ShardedGATLayer(sg, Wm, a, b)(h)
        h.backward(go[sg.own])
        return xl.grad
    res = ls.run(run)
d3 = torch.empty_like(d1)
for r, sg in enumerate(sgs):
    d3[sg.own] = res[r]
deg = torch.bincount(ei[1], minlength=N)
scale = d1.abs().max()
for name, a in (("permuted", d2), ("sharded", d3)):
    e = (a - d1).abs().max(1)[0] / scale
    top = torch.topk(e, 5).indices
    print(name, "max %.1e  L2 %.1e  L2 light %.1e  L2 hub %.1e  rows>1e-4: %d (hub %d)" % (
        float(e.max()), float((a - d1).double().norm() / d1.double().norm()), float((a - d1)[~hub].double().norm() / d1[~hub].double().norm()),
        float((a - d1)[hub].double().norm() / d1[hub].double().norm()), int((e > 1e-4).sum()), int(((e > 1e-4) & hub).sum())),
        [(int(i), bool(hub[i]), int(deg[i]), f"{float(e[i]):.1e}", f"|dX row| {float(d1[i].abs().max() / scale):.1e}") for i in top])
