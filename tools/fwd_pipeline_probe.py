#!/usr/bin/env python3
"""Probe: forward aggregation -> projection pipelined over row chunks on two alternating streams
(chunk c's GEMM resident while chunk c+1 aggregates).  Sub-CSRs are views of the full CSR; no kernel changes.
Measured at C4 (baseline 3.14-3.24 ms for segsum + GEMM): K=2 3.14-3.25 ms, K=4 3.31-3.33 ms, K=8 3.55-3.69 ms,
with the GEMM on 256 / 192 / 128 workgroups -- the chunk tails and the GEMM's late start eat the overlap.  Not taken."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import npi_gnn_amd as npi  # noqa: E402
from npi_gnn_amd import functional as NF  # noqa: E402
from npi_gnn_amd._lib import check, load, ptr  # noqa: E402
from npi_gnn_amd.graph import CSRSide  # noqa: E402
from npi_gnn_amd.synth import bipartite_edge_index  # noqa: E402

N, E, F = 1_000_000, 20_000_000, 256


def sub_sides(side, K, by="entries"):
    lib = load()
    rowptr = side.rowptr.long()
    nnz = int(rowptr[-1])
    if by == "entries":
        cuts = torch.searchsorted(rowptr, torch.arange(1, K, device=rowptr.device) * (nnz // K)).tolist()
    else:
        cuts = [side.n_rows * c // K for c in range(1, K)]
    bounds = [0] + cuts + [side.n_rows]
    out = []
    for r0, r1 in zip(bounds[:-1], bounds[1:]):
        s0, s1 = int(rowptr[r0]), int(rowptr[r1])
        rp = (rowptr[r0:r1 + 1] - s0).to(torch.int32).contiguous()
        nnz_c = s1 - s0
        item = int(lib.npi_item_edges(nnz_c))
        n_items = int(lib.npi_num_items(nnz_c))
        k = torch.arange(0, n_items + 1, device=rp.device, dtype=torch.int64) * item
        ir = torch.searchsorted(rp.long(), k, right=True) - 1
        ir[k >= nnz_c] = r1 - r0
        ir[0] = 0
        sd = CSRSide(rp, side.col[s0:s1], None, None, ir.to(torch.int32).contiguous(), side.status, nnz_c, n_items)
        sd.n_rows, sd.n_cols = r1 - r0, side.n_cols
        out.append((r0, r1, sd))
    return out


def main():
    K = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    by = sys.argv[2] if len(sys.argv) > 2 else "entries"
    dev = torch.device("cuda:0")
    ei = bipartite_edge_index(N, E, seed=20260310).to(dev)
    graph = npi.CSRGraph(ei, N)
    side = graph.by_dst
    x = torch.randn(N, F, device=dev)
    W = torch.randn(F, F, device=dev) / 16
    b = torch.randn(F, device=dev)
    lib = load()

    def gemm(a, out):
        st = torch.cuda.current_stream(dev).cuda_stream
        check(lib.npi_linear_fwd_t(ptr(a), a.stride(0), ptr(W), W.stride(0), ptr(b), None, ptr(out), out.stride(0),
                                   a.size(0), F, F, 0, 0, st), "npi_linear_fwd")

    agg = torch.empty(N, F, device=dev)
    out = torch.empty(N, F, device=dev)

    def baseline():
        NF.segsum(graph, side, x, mean=True, out=agg)
        gemm(agg, out)

    chunks = sub_sides(side, K, by)
    print("chunks (rows, entries):", [(r1 - r0, sd.nnz_max) for r0, r1, sd in chunks])
    sA, sB = torch.cuda.Stream(dev), torch.cuda.Stream(dev)

    def pipelined():
        main_s = torch.cuda.current_stream(dev)
        start = torch.cuda.Event()
        start.record(main_s)
        prev_seg = start
        streams = [sA, sB]
        last = []
        for c, (r0, r1, sd) in enumerate(chunks):
            s = streams[c % 2]
            s.wait_event(prev_seg)                       # seg(c) starts when seg(c-1) is done (and after this stream's GEMM(c-2))
            with torch.cuda.stream(s):
                NF.segsum(None, sd, x, mean=True, out=agg[r0:r1])
                ev = torch.cuda.Event()
                ev.record(s)
                gemm(agg[r0:r1], out[r0:r1])
            prev_seg = ev
        main_s.wait_stream(sA)
        main_s.wait_stream(sB)

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

    baseline()
    ref = out.clone()
    out.zero_()
    pipelined()
    torch.cuda.synchronize()
    print("max |pipelined - baseline| =", float((out - ref).abs().max()))
    print(f"baseline  {timeit(baseline):.3f} ms")
    print(f"pipelined {timeit(pipelined):.3f} ms  (K={K}, by {by})")


if __name__ == "__main__":
    main()
