#!/usr/bin/env python3
"""The C4 layer (SAGEConv 256 -> 256 fwd+bwd, N = 1M, E = 20M) with bf16 STORAGE (f32 accumulation inside the kernels) next to the
f32 layer: ms per step and the deviation of the bf16 result from the f32 one.  Information only: the metric's precision is f32."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import npi_gnn_amd as npi
from npi_gnn_amd.synth import bipartite_edge_index

dev = torch.device("cuda:0")
N, E, F = 1_000_000, 20_000_000, 256
ei = bipartite_edge_index(N, E, seed=20260310).to(dev)
g = npi.CSRGraph(ei, N); _ = g.by_src; del ei
gen = torch.Generator(device=dev).manual_seed(1)
x32 = torch.randn(N, F, device=dev, generator=gen)
go32 = torch.randn(N, F, device=dev, generator=gen)
conv32 = npi.SAGEConv(F, F).to(dev)
res = {}
for name, dt in (("f32", torch.float32), ("bf16", torch.bfloat16)):
    conv = npi.SAGEConv(F, F).to(dev)
    conv.load_state_dict(conv32.state_dict())
    conv = conv.to(dt)
    x = x32.detach().to(dt).clone().requires_grad_(True)
    go = go32.to(dt)
    def step():
        conv.weight.grad = conv.bias.grad = x.grad = None
        conv(x, g).backward(go)
    for _ in range(3): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 20 * 1e3
    out = conv(x, g).detach().float()
    res[name] = (ms, out, x.grad.float(), conv.weight.grad.float())
    print(f"{name}: {ms:.3f} ms per step = {E / ms * 1e3 / 1e9:.2f} G edges/s")
for k, nm in ((1, "out"), (2, "dx"), (3, "dW")):
    a, b = res["f32"][k], res["bf16"][k]
    print(f"bf16 vs f32 {nm}: max |d| / max |ref| = {float((a - b).abs().max() / a.abs().max()):.2e}")
