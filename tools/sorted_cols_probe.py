#!/usr/bin/env python3
"""Does the ORDER of the entries inside a CSR row matter?  The build is a stable sort by row, so a row's columns come in edge-list
order (shuffled for the synthetic graphs).  Here the same graph from the edge list sorted by (target, source): every row's
columns ascend, on both sides.  One SAGEConv and one GATConv layer 256 -> 256, fwd + bwd.
usage: tools/sorted_cols_probe.py [nodes edges [steps]]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import npi_gnn_amd as npi
from npi_gnn_amd.synth import bipartite_edge_index
N, E = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1_000_000, 20_000_000)
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
dev, F = torch.device("cuda:0"), 256
g = torch.Generator().manual_seed(1)
ei = bipartite_edge_index(N, E, seed=20260310).to(dev)
order = torch.argsort(ei[0], stable=True)
ei_s = ei[:, order]
ei_s = ei_s[:, torch.argsort(ei_s[1], stable=True)]          # by target, sources ascending inside a target
x = torch.randn(N, F, generator=g).to(dev).requires_grad_(True)
W = (torch.randn(F, F, generator=g) / 16).to(dev).requires_grad_(True)
att = (torch.randn(1, 1, 2 * F, generator=g) * 0.1).to(dev).requires_grad_(True)
b = torch.zeros(F, device=dev, requires_grad=True)
go = torch.randn(N, F, generator=g).to(dev)
graphs = {}
for name, e in (("shuffled", ei), ("sorted", ei_s)):
    graphs[name] = npi.CSRGraph(e, N); _ = graphs[name].by_src


def run(kind, graph, n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        x.grad = W.grad = att.grad = b.grad = None
        out = npi.sage_conv(x, graph, W, b) if kind == "sage" else npi.gat_conv(x, graph, W, att, b, heads=1, relu=True)
        out.backward(go)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, out.detach()


for kind in ("sage", "gat"):
    res, outs = {k: [] for k in graphs}, {}
    for k in graphs:
        run(kind, graphs[k], 3)
    for _ in range(3):
        for k in graphs:
            t, outs[k] = run(kind, graphs[k], steps)
            res[k].append(t)
    err = float((outs["sorted"] - outs["shuffled"]).abs().max() / outs["shuffled"].abs().max())
    print(f"{kind} N={N} E={E}: shuffled {min(res['shuffled']):.3f} ms, sorted {min(res['sorted']):.3f} ms   "
          f"(best of 3 x {steps}; outputs differ by {err:.1e} relative: another order of the same sums)")
