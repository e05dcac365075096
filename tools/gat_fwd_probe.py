#!/usr/bin/env python3
"""The two forms of the one-head GATConv forward aggregation at the C4 shape, kernel time by HIP events (alternated):
npi_gat_softmax_stats_ex + npi_gat_aggregate_scores against npi_gat_aggregate_fused.  usage: tools/gat_fwd_probe.py [nodes edges]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import npi_gnn_amd as npi
from npi_gnn_amd import functional as NF
from npi_gnn_amd.synth import bipartite_edge_index
N, E = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1_000_000, 20_000_000)
dev, C = torch.device("cuda:0"), 256
g = torch.Generator().manual_seed(1)
graph = npi.CSRGraph(bipartite_edge_index(N, E, seed=20260310).to(dev), N, sort_columns=True)
d = graph.by_dst
h = torch.randn(N, C, generator=g).to(dev)
att = (torch.randn(1, 2 * C, generator=g) / C ** 0.5).to(dev)
a_dst, a_src = NF.gat_scores(h, att, 1, C)


def two_pass():
    m, s, sc = NF.gat_softmax_stats(d, a_dst, a_src, 1, 0.2, want_scores=True)
    return NF.gat_aggregate_scores(d, h, None, C, sc, m, s)


def fused():
    return NF.gat_aggregate_fused(d, h, None, C, a_dst, att, 0.2)[0]


def t(fn, n=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def plain():                                                        # the same gathers without any weight: SAGEConv's aggregation
    return NF.segsum(graph, d, h, mean=True)


res = {"two_pass": [], "fused": [], "plain": []}
for _ in range(3):
    res["two_pass"].append(t(two_pass))
    res["fused"].append(t(fused))
    res["plain"].append(t(plain))
print(f"{os.path.basename(os.environ.get('NPI_GNN_LIB', 'default'))} N={N} E={E}: statistics pass + aggregation {min(res['two_pass']):.3f} ms, "
      f"fused {min(res['fused']):.3f} ms, unweighted mean {min(res['plain']):.3f} ms  ({res})")

# the same launch BEHIND the projection that writes its table (as in the layer): events around the aggregation only
x = torch.randn(N, C, generator=g).to(dev)
W = (torch.randn(C, C, generator=g) / 16).to(dev)
xs = NF.row_scales(x)
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
for k in range(23):
    hh, ad, as_ = NF.linear_fwd_scores(x, W, att.view(-1), a_scales=xs)
    if k >= 3:
        ev[k - 3][0].record()
    o = NF.gat_aggregate_fused(d, hh, None, C, ad, att, 0.2)[0]
    if k >= 3:
        ev[k - 3][1].record()
torch.cuda.synchronize()
ts = sorted(a.elapsed_time(b) for a, b in ev)
print(f"behind the scores GEMM: fused aggregation {ts[0]:.3f} (min) {ts[len(ts) // 2]:.3f} (median) ms")
