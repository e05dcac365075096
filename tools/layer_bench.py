#!/usr/bin/env python3
"""One SAGEConv layer fwd+bwd on the C4 graph in f32 and in bf16 storage (f32 accumulation): the headline metric's
workload at the two storage types.  usage: python tools/layer_bench.py [--hidden 256]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import npi_gnn_amd as npi  # noqa: E402
from npi_gnn_amd.synth import bipartite_edge_index  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--hidden", type=int, default=256)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    N, E, F = 1_000_000, 20_000_000, a.hidden
    g = npi.CSRGraph(bipartite_edge_index(N, E).to(dev), N)
    _ = g.by_src
    for dt in (torch.float32, torch.bfloat16):
        conv = npi.SAGEConv(F, F).to(dev).to(dt)
        x = torch.randn(N, F, device=dev).to(dt).requires_grad_(True)
        go = torch.randn(N, F, device=dev).to(dt)

        def step():
            conv.zero_grad()
            x.grad = None
            conv(x, g).backward(go)
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10):
            step()
        e.record()
        torch.cuda.synchronize()
        t = s.elapsed_time(e) / 10
        print(f"{str(dt):16s} SAGEConv {F}->{F} fwd+bwd, N=1M E=20M: {t:.3f} ms = {E / t / 1e6:.2f} G edges/s", flush=True)


if __name__ == "__main__":
    main()
