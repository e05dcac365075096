#!/usr/bin/env python3
"""Enclosing-subgraph extraction + collate: device kernels vs the CPU restatement of the reference's
Python loops (src/classes.py:652-733), per batch of B target pairs.
usage: python tools/extract_bench.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from npi_gnn_amd.subgraph import InteractionGraph  # noqa: E402
from npi_gnn_amd.synth import bipartite_edge_index  # noqa: E402
from oracle import ref_subgraph as RS  # noqa: E402


def case(num_nodes, pairs_n, F, B, seed=0):
    ei = bipartite_edge_index(num_nodes, 2 * pairs_n, seed=seed)
    n_rna = num_nodes - max(1, num_nodes // 10)
    fwd = ei[:, ei[0] < n_rna]                      # one direction: (rna, protein)
    pairs = fwd.t().contiguous()
    g = torch.Generator().manual_seed(seed)
    usable = torch.rand(pairs.size(0), generator=g) > 0.2
    feat = torch.randn(num_nodes, F, generator=g)
    keys = pairs[torch.randint(0, pairs.size(0), (B,), generator=g)]
    return pairs, usable, feat, keys


def main():
    dev = torch.device("cuda:0")
    for name, (N, P, B, cpu) in {"NPInter2-size graph (5,085 nodes, 20,824 pairs), B=200": (5085, 20824, 200, True),
                                 "same, B=4,166 (a whole test fold)": (5085, 20824, 4166, True),
                                 "1M nodes / 10M pairs, B=200": (1_000_000, 10_000_000, 200, False),
                                 }.items():
        pairs, usable, feat, keys = case(N, P, 177, B)
        t0 = time.perf_counter()
        ig = InteractionGraph(pairs.to(dev), usable.to(dev), feat.to(dev))
        torch.cuda.synchronize()
        t_build = (time.perf_counter() - t0) * 1e3
        kd = keys.to(dev)
        for _ in range(3):
            x, ei, b = ig.batch(kd)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            x, ei, b = ig.batch(kd)
        torch.cuda.synchronize()
        t_gpu = (time.perf_counter() - t0) / 20 * 1e3
        line = f"{name}: n={x.size(0)} e={ei.size(1)}  GPU {t_gpu:.3f} ms per batch (graph upload+CSR once: {t_build:.1f} ms)"
        if cpu:
            adj = RS.adjacency(pairs.tolist(), usable.tolist())           # the reference keeps these lists in memory
            t0 = time.perf_counter()
            ox, oe, ob, on = RS.enclosing_subgraph_batch(pairs, usable, feat, keys)
            t_cpu = (time.perf_counter() - t0) * 1e3
            assert torch.equal(ox, x.cpu()) and torch.equal(oe, ei.cpu())
            line += f"   CPU restatement {t_cpu:.1f} ms (incl. its adjacency build)"
        print(line, flush=True)


if __name__ == "__main__":
    main()
