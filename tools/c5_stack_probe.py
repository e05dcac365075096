#!/usr/bin/env python3
"""The C5 stack (3 x GATConv 256, one head, ReLU fused, N = 4M / E = 100M) on one GPU: ms per step; under rocprofv3
--kernel-trace the kernels of a step.  usage: tools/c5_stack_probe.py [steps]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench_extras as B
import npi_gnn_amd as npi
from npi_gnn_amd.synth import bipartite_edge_index
dev = torch.device("cuda:0")
N5, E5, F5 = 4_000_000, 100_000_000, 256
gen = torch.Generator().manual_seed(11)
g5 = npi.CSRGraph(bipartite_edge_index(N5, E5, seed=2).to(dev), N5); _ = g5.by_src
weights = [((torch.randn(F5, F5, generator=gen) / 16), torch.zeros(F5)) for _ in range(3)]
att = [torch.randn(1, 1, 2 * F5, generator=gen) * 0.1 for _ in range(3)]
x5 = torch.randn(N5, F5, generator=gen).to(dev)
st = B._stack_step("gat", weights, x5, g5, att=att)
st(); torch.cuda.synchronize()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
t0 = time.perf_counter()
for _ in range(n): st()
torch.cuda.synchronize()
print(f"C5 stack: {(time.perf_counter() - t0) / n * 1e3:.2f} ms per step")
