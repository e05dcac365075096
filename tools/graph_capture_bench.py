#!/usr/bin/env python3
"""Full-batch training step on a FIXED graph (BASELINE configs C1-C3: the bundled NPInter2 graph, 5,085 nodes,
41,648 edges) captured into one HIP graph: a handful of microsecond kernels per layer are launch-bound, and
every shape is static, so the whole forward + backward replays as a single graph launch.
usage: python tools/graph_capture_bench.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import npi_gnn_amd as npi  # noqa: E402

G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def main():
    dev = torch.device("cuda:0")
    fx = torch.load(os.path.join(G, "npinter2_graph.pt"), map_location="cpu", weights_only=False)
    x = fx["x"].to(dev)
    ei = fx["edge_index"].long().to(dev)
    N = x.size(0)
    graph = npi.CSRGraph(ei, N)
    _ = graph.by_src
    torch.manual_seed(0)
    convs = torch.nn.ModuleList([npi.SAGEConv(178, 128), npi.SAGEConv(128, 128), npi.SAGEConv(128, 128)]).to(dev)
    target = torch.randn(N, 128, device=dev)

    def step():
        for p in convs.parameters():
            p.grad = None
        h = x
        for c in convs:
            h = torch.relu(c(h, graph))
        loss = (h - target).pow(2).mean()
        loss.backward()
        return loss

    def timeit(fn, n=200):
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    t_eager = timeit(step)
    ref_loss = float(step().detach())
    ref_grads = [p.grad.clone() for p in convs.parameters()]
    # capture (warm-up on a side stream first, as torch.cuda.graphs requires)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            step()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    for p in convs.parameters():
        p.grad = None
    with torch.cuda.graph(g):
        static_loss = step()
    g.replay()
    torch.cuda.synchronize()
    grads = [p.grad for p in convs.parameters()]
    same = all(torch.equal(a, b) for a, b in zip(grads, ref_grads)) and float(static_loss.detach()) == ref_loss
    t_graph = timeit(g.replay)
    print(f"3-layer SAGEConv 178->128->128->128 full-batch fwd+bwd on NPInter2 (N={N}, E={ei.size(1)}): "
          f"eager {t_eager:.3f} ms, HIP graph replay {t_graph:.3f} ms ({t_eager / t_graph:.1f}x), "
          f"bitwise identical loss and gradients: {same}")


if __name__ == "__main__":
    main()
