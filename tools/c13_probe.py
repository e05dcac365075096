#!/usr/bin/env python3
"""Configs 1 and 3 (full-batch GCNConv stacks on the NPInter2 / RPI7317 graphs): ms per step; under rocprofv3 --kernel-trace the
kernels of a step.  usage: tools/c13_probe.py [c1|c3]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench_extras as B
import npi_gnn_amd as npi
from npi_gnn_amd import functional as NF
dev = torch.device("cuda:0")
G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
which = sys.argv[1] if len(sys.argv) > 1 else "c3"
fx = torch.load(os.path.join(G, "npinter2_graph.pt" if which == "c1" else "rpi7317_graph.pt"), map_location="cpu", weights_only=False)
x, ei = fx["x"], fx["edge_index"].long()
g = npi.CSRGraph(ei.to(dev), x.size(0)); _ = g.by_src
norm = NF.GCNNorm(g)
st = B._stack_step("gcn", fx["gcn64" if which == "c1" else "gcn256"], x.to(dev), g, norm=norm)
for _ in range(10): st()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(100): st()
torch.cuda.synchronize()
print(f"{which}: {(time.perf_counter() - t0) / 100 * 1e3:.3f} ms per step (eager)")
