#!/usr/bin/env python3
"""C2 (NPInter2 graph, 3 x SAGEConv 178->128->128->128) full-batch step in bf16 storage and in f32: ms per step; under
rocprofv3 --kernel-trace --stats the kernels of either.  usage: tools/c2_probe.py [bf16|f32]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import npi_gnn_amd as npi
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from config_bench import stack_step, G
dev = torch.device("cuda:0")
fx = torch.load(os.path.join(G, "npinter2_graph.pt"), map_location="cpu", weights_only=False)
x, ei = fx["x"], fx["edge_index"].long()
graph = npi.CSRGraph(ei.to(dev), x.size(0)); _ = graph.by_src
for name in (sys.argv[1:] or ["bf16", "f32"]):
    dt = torch.bfloat16 if name == "bf16" else torch.float32
    st = stack_step("sage", fx["sage_weights"], x.to(dev), graph, dtype=dt)
    for _ in range(10): st()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(100): st()
    torch.cuda.synchronize()
    print(f"C2 {name}: {(time.perf_counter() - t0) / 100 * 1e3:.3f} ms per step")
