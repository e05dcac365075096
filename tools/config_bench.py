#!/usr/bin/env python3
"""The five BASELINE.json configs on ONE MI355X: parity against the oracle (C1-C3, on the bundled NPInter2
graph of tests/golden/npinter2_graph.pt) and time per full-batch step (forward + backward over the layer
stack).  C4 is bench.py's workload (one layer), C5 is run with one GAT layer per call three times over on one
GPU (the 8-GPU form of C4/C5 is bench.py --gpus 8).
usage: python tools/config_bench.py [--skip-c5]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import npi_gnn_amd as npi  # noqa: E402
from npi_gnn_amd import functional as NF  # noqa: E402
from npi_gnn_amd.synth import bipartite_edge_index  # noqa: E402
from oracle import ref_conv as R  # noqa: E402

G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def timeit(fn, n=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def stack_step(kind, weights, x, graph, dtype=torch.float32, norm=None, att=None):
    """forward + backward through a stack of convs with relu between them; returns the step function"""
    dev = x.device
    params = [(W.to(dev).to(dtype).requires_grad_(True), b.to(dev).to(dtype).requires_grad_(True)) for W, b in weights]
    atts = [a.to(dev).requires_grad_(True) for a in att] if att else None
    xin = x.to(dtype).requires_grad_(True)

    def step():
        for W, b in params:
            W.grad = b.grad = None
        xin.grad = None
        h = xin
        for k, (W, b) in enumerate(params):
            if kind == "sage":
                h = npi.sage_conv(h, graph, W, b)
            elif kind == "gcn":
                h = NF.gcn_conv(h, None, W, b, norm=norm)
            else:
                h = npi.gat_conv(h, graph, W, atts[k], b, heads=1)
            h = torch.relu(h)
        h.float().pow(2).mean().backward()
        return h
    return step


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--skip-c5", action="store_true")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    fx = torch.load(os.path.join(G, "npinter2_graph.pt"), map_location="cpu", weights_only=False)
    x, ei = fx["x"], fx["edge_index"].long()
    N, E = x.size(0), ei.size(1)
    graph = npi.CSRGraph(ei.to(dev), N)
    _ = graph.by_src
    rows = fx["rows"]
    print(f"| config | workload | parity vs oracle | time per step (fwd+bwd) |\n|---|---|---|---|")

    # C1: 2-layer GCN hidden 64 (the CPU-plumbing config; run on the GPU as well)
    norm = NF.GCNNorm(graph)
    with torch.no_grad():
        h = x.to(dev)
        for W, b in fx["gcn64"]:
            h = torch.relu(NF.gcn_conv(h, None, W.to(dev), b.to(dev), norm=norm))
    err = float((h.cpu()[rows] - fx["gcn64_out"]).abs().max())
    t = timeit(stack_step("gcn", fx["gcn64"], x.to(dev), graph, norm=norm))
    print(f"| C1 | NPInter2 graph N={N} E={E}, 2 x GCNConv 178->64->64 fp32 | max abs err {err:.1e} | {t:.3f} ms |")

    # C2: 3-layer SAGE hidden 128, bf16 storage
    with torch.no_grad():
        h = x.to(dev).to(torch.bfloat16)
        for W, b in fx["sage_weights"]:
            h = torch.relu(npi.sage_conv(h, graph, W.to(dev).to(torch.bfloat16), b.to(dev).to(torch.bfloat16)))
    ref = fx["sage3_out"]
    err = float((h.float().cpu()[rows] - ref).abs().max() / ref.abs().max())
    t = timeit(stack_step("sage", fx["sage_weights"], x.to(dev), graph, dtype=torch.bfloat16))
    t32 = timeit(stack_step("sage", fx["sage_weights"], x.to(dev), graph))
    print(f"| C2 | same graph, 3 x SAGEConv 178->128->128->128, bf16 storage / f32 accumulate | max err {err:.1e} of max (bf16) | "
          f"{t:.3f} ms (fp32: {t32:.3f} ms) |")

    # C3: 3-layer GCN hidden 256 fp32
    with torch.no_grad():
        h = x.to(dev)
        for W, b in fx["gcn256"]:
            h = torch.relu(NF.gcn_conv(h, None, W.to(dev), b.to(dev), norm=norm))
    err = float((h.cpu()[rows] - fx["gcn256_out"]).abs().max())
    t = timeit(stack_step("gcn", fx["gcn256"], x.to(dev), graph, norm=norm))
    print(f"| C3 | same graph (RPI7317-scale: launch/LLC-bound), 3 x GCNConv 178->256->256->256 fp32 | max abs err {err:.1e} | {t:.3f} ms |")

    # C4: pointer to bench.py
    print("| C4 | synthetic bipartite N=1M E=20M, 1 x SAGEConv 256->256 fp32 | tests/test_gpu_parity.py (1e-4) | bench.py: see profiles/r01e_bench.json |")

    if not a.skip_c5:
        N5, E5, F5 = 4_000_000, 100_000_000, 256
        ei5 = bipartite_edge_index(N5, E5, seed=2).to(dev)
        g5 = npi.CSRGraph(ei5, N5)
        _ = g5.by_src
        del ei5
        gen = torch.Generator().manual_seed(0)
        weights = [((torch.randn(F5, F5, generator=gen) / 16), torch.zeros(F5)) for _ in range(3)]
        att = [torch.randn(1, 1, 2 * F5, generator=gen) * 0.1 for _ in range(3)]
        x5 = torch.randn(N5, F5, generator=gen).to(dev)
        t = timeit(stack_step("gat", weights, x5, g5, att=att), n=3, warm=1)
        print(f"| C5 | synthetic bipartite N=4M E=100M, 3 x GATConv 256 (1 head), fp32, ONE GPU | tests/test_gpu_gat.py (1e-4) | "
              f"{t:.1f} ms = {3 * E5 / t / 1e6:.2f} G edge-layers/s |")


if __name__ == "__main__":
    main()
