#!/usr/bin/env python3
"""One SAGEConv layer 256 -> 256, forward + backward at the C4 size, with the projection GEMMs on six bf16 products (default up to
round 5), the forward one on three fp16 products (NPI_GEMM_SPLIT_F16X2; round 5), or both (round 6: the backward aggregate-first), the row scales of the aggregate written by the aggregation launch or by
a pass of their own.  Variant libraries through NPI_GNN_LIB.  usage: tools/f16x2_layer_probe.py [--separate-scales]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import npi_gnn_amd as npi
from npi_gnn_amd import functional as NF
from npi_gnn_amd.synth import bipartite_edge_index_device

dev = torch.device("cuda:0")
N, E, F = 1_000_000, 20_000_000, 256
graph = npi.CSRGraph(bipartite_edge_index_device(N, E, dev), N, sort_columns=True)
_ = graph.by_src
torch.manual_seed(0)
x = torch.randn(N, F, device=dev).requires_grad_(True)
go = torch.randn(N, F, device=dev)
conv = npi.SAGEConv(F, F).to(dev)
if "--separate-scales" in sys.argv:
    NF.segsum_scales_ok = lambda *a, **k: False


def step():
    conv.zero_grad(); x.grad = None
    out = conv(x, graph)
    out.backward(go)
    return out


def t(n=20):
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        step()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


from npi_gnn_amd.schedule import DEFAULT
VARIANTS = {"bf16x3": DEFAULT.but(f16x2_min_rows=None), "fp16x2": DEFAULT.but(aggregate_first_backward=False),
            "fp16x2+aggfirst": DEFAULT}
res = {}
for name in ("bf16x3", "fp16x2", "fp16x2+aggfirst", "bf16x3 again", "fp16x2 again", "fp16x2+aggfirst again"):
    conv.schedule = VARIANTS[name.split()[0]]
    ms = t()
    out = step().detach()
    res[name] = (ms, out, x.grad.clone(), conv.weight.grad.clone())
    print(f"{name}: {ms:.3f} ms / step", flush=True)
a, b = res["bf16x3"], res["fp16x2+aggfirst"]
idx = torch.arange(0, N, 997, device=dev)[:1000]
print("out   max |fp16x2 - bf16x3| / max:", float((a[1] - b[1]).abs().max() / a[1].abs().max()))
print("dX    max diff / max:", float((a[2] - b[2]).abs().max() / a[2].abs().max()))
print("dW    max diff / max:", float((a[3] - b[3]).abs().max() / a[3].abs().max()))
print("lib:", os.environ.get("NPI_GNN_LIB", "default"))
