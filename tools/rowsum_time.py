#!/usr/bin/env python3
"""HIP-event time of the per-row sums of a per-entry scalar alone (GATConv's g_src / g_dst): plain (coalesced) and through the
transpose map (a 4-byte gather per entry), on the synthetic bipartite graph -- for same-box A/B runs of variant libraries
(NPI_GNN_LIB).  usage: tools/rowsum_time.py [nodes edges]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import npi_gnn_amd as npi
from npi_gnn_amd import functional as NF
from npi_gnn_amd.synth import bipartite_edge_index_device
dev = torch.device("cuda:0")
N, E = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1_000_000, 20_000_000)
def stage(m):
    torch.cuda.synchronize(); print("  ..", m, file=sys.stderr, flush=True)


stage("start")
graph = npi.CSRGraph(bipartite_edge_index_device(N, E, dev, seed=2), N, sort_columns=True)
sr, d = graph.by_src, graph.by_dst
stage("graph")
tmap = NF._inverse_transpose_map(graph)
stage("map")
dz = torch.randn(max(sr.nnz_max, 1), 1, device=dev)
res = []
for name, fn in (("plain", lambda: NF.seg_rowsum(sr, dz, 1)), ("mapped", lambda: NF.seg_rowsum(d, dz, 1, map_=tmap))):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn()
    e1.record()
    torch.cuda.synchronize()
    res.append(f"{name} {e0.elapsed_time(e1) / 20:.4f} ms")
    stage(res[-1])
# (row sums from a running sum in fp64: index_add_ on the hub rows is millions of atomics on one address -- minutes at the C5 size)
nnz = int(d.rowptr[-1])
cs = torch.cat([torch.zeros(1, dtype=torch.float64, device=dev), dz[tmap[:nnz].long(), 0].double().cumsum(0)])
want = cs[d.rowptr[1:].long()] - cs[d.rowptr[:-1].long()]
err = float((NF.seg_rowsum(d, dz, 1, map_=tmap).view(-1).double() - want).abs().max())
print(os.path.basename(os.environ.get("NPI_GNN_LIB", "default")), N, E, "|", "  ".join(res), f"| mapped max err {err:.1e}")
