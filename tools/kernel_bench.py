#!/usr/bin/env python3
"""Per-entry-point timing on one MI355X (HIP events, interleaved rounds in one process).
usage: python tools/kernel_bench.py [--gemm] [--seg] [--nodes N --edges E --hidden F] [--rounds R]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import npi_gnn_amd as npi  # noqa: E402
from npi_gnn_amd import functional as NF  # noqa: E402


def timeit(fn, rounds):
    ts = []
    for _ in range(rounds):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        fn()
        e.record()
        torch.cuda.synchronize()
        ts.append(s.elapsed_time(e))
    ts = ts[1:] if len(ts) > 1 else ts
    ts.sort()
    return ts[len(ts) // 2], ts[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gemm", action="store_true")
    ap.add_argument("--seg", action="store_true")
    ap.add_argument("--nodes", type=int, default=1_000_000)
    ap.add_argument("--edges", type=int, default=20_000_000)
    ap.add_argument("--hidden", type=int, default=256)
    ap.add_argument("--rounds", type=int, default=8)
    ap.add_argument("--gat", action="store_true")
    ap.add_argument("--heads", type=int, default=1)
    a = ap.parse_args()
    if not (a.gemm or a.seg):
        a.gemm = a.seg = True
    dev = torch.device("cuda:0")
    N, E, F = a.nodes, a.edges, a.hidden
    g = torch.Generator().manual_seed(0)
    if a.gemm:
        A = torch.randn(N, F, generator=g).to(dev)
        dC = torch.randn(N, F, generator=g).to(dev)
        W = (torch.randn(F, F, generator=g) / F ** 0.5).to(dev)
        b = torch.randn(F, generator=g).to(dev)
        rs = torch.rand(N, generator=g).to(dev)
        fl = 2.0 * N * F * F
        for name, fn in (("linear_fwd", lambda: NF.linear_fwd(A, W, b)),
                         ("linear_bwd_data", lambda: NF.linear_bwd_data(dC, W, rs)),
                         ("linear_bwd_weight(+db)", lambda: NF.linear_bwd_weight(A, dC, True)),
                         ("colsum", lambda: NF.colsum(dC))):
            med, best = timeit(fn, a.rounds)
            print(f"{name:26s} median {med:8.3f} ms  min {best:8.3f} ms  {fl / (med * 1e-3) / 1e12:7.1f} TF/s(median)", flush=True)
        Ab, dCb, Wb, bb = A.to(torch.bfloat16), dC.to(torch.bfloat16), W.to(torch.bfloat16), b.to(torch.bfloat16)
        for name, fn in (("linear_fwd bf16", lambda: NF.linear_fwd(Ab, Wb, bb)),
                         ("linear_bwd_data bf16", lambda: NF.linear_bwd_data(dCb, Wb, rs)),
                         ("linear_bwd_weight bf16", lambda: NF.linear_bwd_weight(Ab, dCb, True))):
            med, best = timeit(fn, a.rounds)
            print(f"{name:26s} median {med:8.3f} ms  min {best:8.3f} ms  {fl / (med * 1e-3) / 1e12:7.1f} TF/s(median)  "
                  f"{(2.0 * N * F * 2) / (med * 1e-3) / 1e9:7.0f} GB/s in+out", flush=True)
        del A, dC, Ab, dCb
    if a.seg:
        from npi_gnn_amd.synth import bipartite_edge_index
        ei = bipartite_edge_index(N, E).to(dev)
        x = torch.randn(N, F, generator=g).to(dev)
        graph = npi.CSRGraph(ei, N)
        _ = graph.by_src
        nbytes = E * (F * 4 + 4) + N * (2 * F * 4 + 4)
        for name, fn in (("segsum fwd (mean)", lambda: NF.segsum(graph, graph.by_dst, x, mean=True)),
                         ("segsum bwd (sum, by_src)", lambda: NF.segsum(graph, graph.by_src, x))):
            med, best = timeit(fn, a.rounds)
            print(f"{name:26s} median {med:8.3f} ms  min {best:8.3f} ms  {nbytes / (med * 1e-3) / 1e9:8.1f} GB/s algorithmic", flush=True)
        med, best = timeit(lambda: npi.CSRGraph(ei, N), a.rounds)
        print(f"{'csr build (by_dst)':26s} median {med:8.3f} ms  min {best:8.3f} ms")
        xb = x.to(torch.bfloat16)
        nbytes_b = E * (F * 2 + 4) + N * (2 * F * 2 + 4)
        med, best = timeit(lambda: NF.segsum(graph, graph.by_dst, xb, mean=True), a.rounds)
        print(f"{'segsum fwd bf16 storage':26s} median {med:8.3f} ms  min {best:8.3f} ms  {nbytes_b / (med * 1e-3) / 1e9:8.1f} GB/s algorithmic", flush=True)
        if a.gat:
            H = a.heads
            C = F // H
            Wg = (torch.randn(F, F, generator=g) / F ** 0.5).to(dev).requires_grad_(True)
            att = (torch.randn(1, H, 2 * C, generator=g) * 0.1).to(dev).requires_grad_(True)
            bg = torch.zeros(F, device=dev, requires_grad=True)
            xg = x.clone().requires_grad_(True)
            go = torch.randn(N, F, generator=g).to(dev)

            def gat_step():
                xg.grad = Wg.grad = att.grad = bg.grad = None
                npi.gat_conv(xg, graph, Wg, att, bg, heads=H).backward(go)
            med, best = timeit(gat_step, max(3, a.rounds // 2))
            print(f"{'GATConv fwd+bwd (H=%d)' % H:26s} median {med:8.3f} ms  min {best:8.3f} ms  {E / (med * 1e-3) / 1e9:6.3f} G edges/s", flush=True)


if __name__ == "__main__":
    main()
