#!/usr/bin/env python3
"""One GATConv 256 -> 256 layer (one head) forward + backward under Schedule alternatives, alternated in one process:
usage: tools/gat_sched_probe.py "<field>=<value>[,...]" [nodes edges [steps]]   (the alternative against the default)"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import npi_gnn_amd as npi
from npi_gnn_amd.schedule import DEFAULT
from npi_gnn_amd.synth import bipartite_edge_index
alt = DEFAULT.but(**eval("dict(" + sys.argv[1] + ")"))
N, E = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1_000_000, 20_000_000)
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 20
dev, F = torch.device("cuda:0"), 256
g = torch.Generator().manual_seed(1)
graph = npi.CSRGraph(bipartite_edge_index(N, E, seed=2).to(dev), N); _ = graph.by_src
x = torch.randn(N, F, generator=g).to(dev).requires_grad_(True)
W = (torch.randn(F, F, generator=g) / 16).to(dev).requires_grad_(True)
att = (torch.randn(1, 1, 2 * F, generator=g) * 0.1).to(dev).requires_grad_(True)
b = torch.zeros(F, device=dev, requires_grad=True)
go = torch.randn(N, F, generator=g).to(dev)


def run(sch, n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        x.grad = W.grad = att.grad = b.grad = None
        npi.gat_conv(x, graph, W, att, b, heads=1, relu=True, schedule=sch).backward(go)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for s in (DEFAULT, alt):
    run(s, 3)
res = {"default": [], "alternative": []}
for _ in range(3):
    res["default"].append(run(DEFAULT, steps))
    res["alternative"].append(run(alt, steps))
print(f"N={N} E={E}: default {min(res['default']):.3f} ms, {sys.argv[1]} {min(res['alternative']):.3f} ms  (best of 3 x {steps} steps; all: {res})")
