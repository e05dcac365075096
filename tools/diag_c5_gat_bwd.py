#!/usr/bin/env python3
"""Diagnostic: which rows of dX differ at the C5 size, and do the big GEMMs agree with torch at 4M rows?"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import npi_gnn_amd as npi
from npi_gnn_amd import functional as NF
from npi_gnn_amd.synth import bipartite_edge_index

dev = torch.device("cuda:0")
N5, E5, F = 4_000_000, 100_000_000, 256
g = torch.Generator().manual_seed(17)
W = ((torch.rand(F, F, generator=g) * 2 - 1) * (6.0 / (2 * F)) ** 0.5).to(dev)
torch.manual_seed(11)
A = torch.randn(N5, F, device=dev)
for name, fn, ref in (("bwd_data", lambda: NF.linear_bwd_data(A, W), lambda: A @ W.t()),
                      ("fwd", lambda: NF.linear_fwd(A, W), lambda: A @ W)):
    got, want = fn(), ref()
    err = (got - want).abs().amax(1)
    bad = torch.nonzero(err > 1e-2).flatten()
    print(name, "max err", float(err.max()), "bad rows", bad.numel(), bad[:10].tolist(), bad[-10:].tolist(), flush=True)
    del got, want
ei = bipartite_edge_index(N5, E5, seed=2).to(dev)
graph = npi.CSRGraph(ei, N5)
x = A
att = ((torch.rand(1, 1, 2 * F, generator=g) * 2 - 1) * (6.0 / (1 + 2 * F)) ** 0.5 * 3.0).to(dev).requires_grad_(True)
b = (torch.randn(F, generator=g) * 0.1).to(dev).requires_grad_(True)
Wg = W.clone().requires_grad_(True)
xg = x.detach().requires_grad_(True)
go = torch.randn(N5, F, device=dev)
out = npi.gat_conv(xg, graph, Wg, att, b, heads=1)
out.backward(go)
dx = xg.grad.detach()
out = out.detach()
with torch.no_grad():
    side = graph.by_dst
    nnz = int(side.rowptr[-1])
    row, col = side.rowidx[:nnz].long(), side.col[:nnz].long()
    h = x @ W
    a_d, a_s = att.detach().view(-1)[:F], att.detach().view(-1)[F:]
    s_dst, s_src = (h.double() @ a_d.double()), (h.double() @ a_s.double())
    pre = s_dst[row] + s_src[col]
    z = torch.nn.functional.leaky_relu(pre, 0.2)
    m = torch.full((N5,), -1e300, dtype=torch.float64, device=dev).scatter_reduce(0, row, z, "amax")
    ez = torch.exp(z - m[row])
    ssum = torch.zeros(N5, dtype=torch.float64, device=dev).index_add_(0, row, ez)
    alpha = ez / (ssum[row] + 1e-16)
    del ez, z
    dot = torch.empty(nnz, dtype=torch.float64, device=dev)
    CH = 4_000_000
    for p0 in range(0, nnz, CH):
        sl = slice(p0, min(p0 + CH, nnz))
        dot[sl] = (go[row[sl]].double() * h[col[sl]].double()).sum(1)
    D = (go.double() * (out.double() - b.detach().double())).sum(1)
    dz = alpha * (dot - D[row]) * torch.where(pre > 0, 1.0, 0.2)
    g_dst = torch.zeros(N5, dtype=torch.float64, device=dev).index_add_(0, row, dz)
    g_src = torch.zeros(N5, dtype=torch.float64, device=dev).index_add_(0, col, dz)
    dh = (g_dst[:, None] * a_d.double()[None, :] + g_src[:, None] * a_s.double()[None, :]).float()
    a32 = alpha.float()
    t0 = time.time()
    for p0 in range(0, nnz, CH):
        sl = slice(p0, min(p0 + CH, nnz))
        dh.index_add_(0, col[sl], a32[sl, None] * go[row[sl]])
    torch.cuda.synchronize()
    print("fp32 index_add", time.time() - t0, "s", flush=True)
    ref_dx = dh @ W.t()
    err = (dx - ref_dx).abs().amax(1)
    scale = ref_dx.abs().amax(1).clamp(min=float(ref_dx.abs().mean()))
    rel = err / scale
    deg_in = (side.rowptr[1:] - side.rowptr[:-1]).long()
    so = graph.by_src
    deg_out = (so.rowptr[1:] - so.rowptr[:-1]).long()
    bad = torch.nonzero(rel > 1e-2).flatten()
    print("rows with rel err > 1e-2:", bad.numel(), flush=True)
    for r in bad[:40].tolist():
        print(r, "rel", float(rel[r]), "in", int(deg_in[r]), "out", int(deg_out[r]), "got", dx[r, :3].tolist(), "ref", ref_dx[r, :3].tolist())
    print("max rel over rows with deg < 1e4:", float(rel[(deg_in < 10000) & (deg_out < 10000)].max()))
