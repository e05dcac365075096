#!/usr/bin/env python3
"""Probe: source-blocked aggregation.  Entries are regrouped into virtual rows (source block, target row) so that
one pass touches only `S` source rows (S KB of x: resident in L2 / Infinity Cache), the virtual-row partials are
then summed per target row by a second segsum.  Existing kernels only; index plumbing in torch.
usage: colblock_probe.py [S ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import npi_gnn_amd as npi  # noqa: E402
from npi_gnn_amd import functional as NF  # noqa: E402
from npi_gnn_amd._lib import load  # noqa: E402
from npi_gnn_amd.graph import CSRSide  # noqa: E402
from npi_gnn_amd.synth import bipartite_edge_index  # noqa: E402

N, E, F = 1_000_000, 20_000_000, 256


def make_side(rowptr, col, n_rows, n_cols):
    lib = load()
    nnz = int(rowptr[-1])
    item = int(lib.npi_item_edges(nnz))
    n_items = int(lib.npi_num_items(nnz))
    k = torch.arange(0, n_items + 1, device=rowptr.device, dtype=torch.int64) * item
    ir = torch.searchsorted(rowptr.long(), k, right=True) - 1
    ir[k >= nnz] = n_rows
    ir[0] = 0
    sd = CSRSide(rowptr.to(torch.int32).contiguous(), col.to(torch.int32).contiguous(), None, None,
                 ir.to(torch.int32).contiguous(), torch.zeros(1, dtype=torch.int32, device=rowptr.device), nnz, n_items)
    sd.n_rows, sd.n_cols = n_rows, n_cols
    return sd


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    dev = torch.device("cuda:0")
    ei = bipartite_edge_index(N, E, seed=20260310).to(dev)
    graph = npi.CSRGraph(ei, N)
    x = torch.randn(N, F, device=dev)
    base = NF.segsum(graph, graph.by_dst, x)
    t_base = timeit(lambda: NF.segsum(graph, graph.by_dst, x))
    print(f"baseline segsum {t_base:.3f} ms")
    loops = torch.arange(N, device=dev)
    src = torch.cat([ei[0], loops])
    dst = torch.cat([ei[1], loops])
    for S in [int(a) for a in sys.argv[1:]] or [65536, 131072, 262144]:
        blk = src // S
        key = blk * N + dst
        key_s, order = torch.sort(key, stable=True)
        col = src[order]
        vkeys, counts = torch.unique_consecutive(key_s, return_counts=True)
        V = vkeys.numel()
        vrowptr = torch.zeros(V + 1, dtype=torch.int64, device=dev)
        vrowptr[1:] = torch.cumsum(counts, 0)
        s1 = make_side(vrowptr, col, V, N)
        # level 2: real row -> its virtual rows
        vreal = vkeys % N
        _, order2 = torch.sort(vreal, stable=True)
        rowptr2 = torch.zeros(N + 1, dtype=torch.int64, device=dev)
        rowptr2[1:] = torch.cumsum(torch.bincount(vreal, minlength=N), 0)
        s2 = make_side(rowptr2, order2, N, V)
        part = torch.empty(V, F, device=dev)
        out = torch.empty(N, F, device=dev)

        def run():
            NF.segsum(None, s1, x, out=part)
            NF.segsum(None, s2, part, out=out)
        run()
        torch.cuda.synchronize()
        err = float((out - base).abs().max())
        t1 = timeit(lambda: NF.segsum(None, s1, x, out=part))
        t2 = timeit(lambda: NF.segsum(None, s2, part, out=out))
        print(f"S={S}: virtual rows {V} ({V / N:.2f} per row)  level1 {t1:.3f} ms  level2 {t2:.3f} ms  "
              f"total {timeit(run):.3f} ms   max |diff| {err:.2e}")


if __name__ == "__main__":
    main()
