#!/usr/bin/env python3
"""CPU-baseline thread sweep on the GPU box's host (oracle SAGE layer fwd+bwd, 1/10-scale C4)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from npi_gnn_amd.synth import bipartite_edge_index  # noqa: E402
from oracle import ref_conv as R  # noqa: E402

N, E, F = 100_000, 2_000_000, 256
ei = bipartite_edge_index(N, E, seed=20260310)
g = torch.Generator().manual_seed(1)
x = torch.randn(N, F, generator=g)
W = (torch.rand(F, F, generator=g) * 2 - 1) / F ** 0.5
b = torch.zeros(F)
go = torch.randn(N, F, generator=g)
print("os.cpu_count()", os.cpu_count())
for nt in (8, 16, 32, 64, 128, 256):
    if nt > (os.cpu_count() or 1):
        break
    torch.set_num_threads(nt)
    ts = []
    for it in range(4):
        t0 = time.time()
        R.sage_layer_fwd_bwd(x, ei, W, b, go)
        ts.append(time.time() - t0)
    print(f"threads {nt:4d}: best {min(ts[1:]) * 1e3:8.1f} ms  = {E / min(ts[1:]) / 1e6:7.2f} M edges/s", flush=True)
