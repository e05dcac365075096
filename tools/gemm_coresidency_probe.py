#!/usr/bin/env python3
"""Upper bound for hiding the two exposed projection GEMMs of the SAGE step under the aggregation: the
forward aggregation and a projection GEMM of the same size run ALONE and then CONCURRENTLY on two HIP streams over INDEPENDENT
buffers (no dependency at all -- no design that tracks row blocks between the two kernels can do better than this).
usage: python tools/gemm_coresidency_probe.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import npi_gnn_amd as npi
from npi_gnn_amd import functional as NF
from npi_gnn_amd.synth import bipartite_edge_index

dev = torch.device("cuda:0")
N, E, F = 1_000_000, 20_000_000, 256
ei = bipartite_edge_index(N, E, seed=20260310).to(dev)
g = npi.CSRGraph(ei, N); del ei
x = torch.randn(N, F, device=dev)
a = torch.randn(N, F, device=dev)
W = torch.randn(F, F, device=dev) / 16
b = torch.randn(F, device=dev)
agg = torch.empty(N, F, device=dev)
side = torch.cuda.Stream(dev)

def seg():
    NF.segsum(g, g.by_dst, x, mean=True, out=agg)
def gemm():
    return NF.linear_fwd(a, W, b)
def both(gemm_first):
    main = torch.cuda.current_stream(dev)
    side.wait_stream(main)
    if gemm_first:
        with torch.cuda.stream(side):
            o = gemm()
        seg()
    else:
        seg()
        with torch.cuda.stream(side):
            o = gemm()
    main.wait_stream(side)
    return o

def t(fn, n=20):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

ts, tg = t(seg), t(gemm)
tb1, tb2 = t(lambda: both(True)), t(lambda: both(False))
print(f"aggregation alone {ts:.3f} ms, projection GEMM alone {tg:.3f} ms, sum {ts + tg:.3f} ms")
print(f"concurrent, GEMM launched first {tb1:.3f} ms, aggregation first {tb2:.3f} ms  => at most {ts + tg - min(tb1, tb2):.3f} ms "
      f"of the GEMM's {tg:.3f} ms can be hidden ({(ts + tg - min(tb1, tb2)) / tg * 100:.0f} %), with no dependency between the two")
