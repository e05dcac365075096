#!/usr/bin/env python3
"""Does a COLUMN-sliced aggregation win L2 residency?  One launch gathers whole 1 KiB rows of a 102 MB hub table (a 4 MB
XCD L2 holds 4 % of it); S launches over column slices of width F / S gather 1 KiB / S pieces of a table slice of 102 / S MB.
Times the plain launch against the S slice launches (same items, same sums) at the C4 shape, forward orientation.
usage: tools/colslice_probe.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import npi_gnn_amd as npi
from npi_gnn_amd import functional as NF
from npi_gnn_amd.synth import bipartite_edge_index
dev = torch.device("cuda:0")
N, E, F = 1_000_000, 20_000_000, 256
g = npi.CSRGraph(bipartite_edge_index(N, E).to(dev), N)
x = torch.randn(N, F, device=dev)
out = torch.empty(N, F, device=dev)


def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for side_name in ("by_dst", "by_src"):
    side = getattr(g, side_name)
    ref = NF.segsum(g, side, x, mean=True).clone()
    print(side_name, "plain: %.3f ms" % timeit(lambda: NF.segsum(g, side, x, mean=True, out=out)))
    for S in (2, 4, 8):
        w = F // S
        def sliced():
            for s in range(S):
                NF.segsum(g, side, x[:, s * w:(s + 1) * w], mean=True, out=out[:, s * w:(s + 1) * w])
        sliced(); torch.cuda.synchronize()
        print("  %d slices of %d columns: %.3f ms   max |diff| %.2e" % (S, w, timeit(sliced), float((out - ref).abs().max())))
