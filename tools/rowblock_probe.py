"""Probe: the forward of a SAGEConv layer at the C4 size with the aggregation cut into ROW BLOCKS, block b's projection GEMM
(MFMA-bound) running on a second stream beside block b+1's aggregation (HBM-bound).  Row blocks are CSR slices: a rebased
``rowptr`` slice, a view of ``col``, an item list of their own -- the existing entry points, nothing new in the library.

    python tools/rowblock_probe.py [--blocks 1,2,4,8] [--balance entries|rows]
"""
from __future__ import annotations

import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import npi_gnn_amd as npi                                                  # noqa: E402
from npi_gnn_amd import functional as F_                                   # noqa: E402
from npi_gnn_amd import synth                                              # noqa: E402
from npi_gnn_amd.graph import CSRSide                                      # noqa: E402


def block_sides(side: CSRSide, bounds):
    rp = side.rowptr
    dev = rp.device
    offs = rp[torch.tensor(bounds, device=dev)].tolist()
    out = []
    for b in range(len(bounds) - 1):
        r0, r1, o0, o1 = bounds[b], bounds[b + 1], offs[b], offs[b + 1]
        rowptr = (rp[r0:r1 + 1] - o0).contiguous()
        nnz = o1 - o0
        n_items = -(-nnz // side.item)
        k = torch.arange(n_items + 1, device=dev, dtype=torch.int64) * side.item
        ir = torch.searchsorted(rowptr.long(), k, right=True) - 1
        ir[0] = 0
        ir[k >= nnz] = r1 - r0
        out.append(CSRSide(rowptr, side.col[o0:o1], side.eid[o0:o1], side.rowidx[o0:o1], ir.int().contiguous(), side.status, nnz,
                           n_items, n_rows=r1 - r0, n_cols=side.n_cols, item=side.item))
    return out


def bounds_for(side: CSRSide, B: int, balance: str):
    N = side.n_rows
    if B == 1:
        return [0, N]
    if balance == "rows":
        return [N * b // B for b in range(B)] + [N]
    rp = side.rowptr.long()
    cost = rp + torch.arange(N + 1, device=rp.device) * 2            # entries + 2 per row (self row read + row write)
    cuts = [0]
    for b in range(1, B):
        cuts.append(int(torch.searchsorted(cost, cost[-1] * b // B)))
    return cuts + [N]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--blocks", default="1,2,4,8")
    ap.add_argument("--balance", default="entries")
    ap.add_argument("--nodes", type=int, default=1_000_000)
    ap.add_argument("--edges", type=int, default=20_000_000)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--reserve", type=int, default=0)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    ei = synth.bipartite_edge_index_device(args.nodes, args.edges, dev)
    N, F = args.nodes, 256
    g = npi.CSRGraph(ei, N)
    side = g.by_dst
    torch.manual_seed(1)
    x = torch.randn(N, F, device=dev)
    W = torch.randn(F, F, device=dev) / 16
    bias = torch.randn(F, device=dev)
    wsf, _ = F_.prepare_weight(W, backward=False)
    main_s = torch.cuda.current_stream(dev)
    side_s = torch.cuda.Stream(device=dev)
    res = {}

    ref_agg = F_.segsum(g, side, x, mean=True)
    ref_out = F_.linear_fwd(ref_agg, W, bias, relu=True, ws=wsf)

    def timeit(fn):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / args.reps

    for B in [int(b) for b in args.blocks.split(",")]:
        bounds = bounds_for(side, B, args.balance)
        sides = block_sides(side, bounds)
        agg = torch.empty(N, F, device=dev)
        out = torch.empty(N, F, device=dev)
        evs = [torch.cuda.Event() for _ in sides]

        two = [main_s, side_s]

        def step_alt():
            # block b on stream b % 2: aggregation, then its GEMM right behind it; block b+1's aggregation (other stream) waits for
            # block b's AGGREGATION only -- the GEMM is dispatched first and is resident when the next aggregation fills the rest
            side_s.wait_stream(main_s)
            prev = None
            for b, sd in enumerate(sides):
                r0, r1 = bounds[b], bounds[b + 1]
                s = two[b % 2]
                with torch.cuda.stream(s):
                    if prev is not None:
                        s.wait_event(prev)
                    F_.segsum(None, sd, x, mean=True, out=agg[r0:r1])
                    evs[b].record(s)
                    prev = evs[b]
                    F_.linear_fwd(agg[r0:r1], W, bias, relu=True, ws=wsf, out=out[r0:r1], reserve_cus=args.reserve)
            main_s.wait_stream(side_s)

        def step(overlap=True):
            if overlap:
                side_s.wait_stream(main_s)
            for b, sd in enumerate(sides):
                r0, r1 = bounds[b], bounds[b + 1]
                F_.segsum(None, sd, x, mean=True, out=agg[r0:r1])
                if overlap:
                    evs[b].record(main_s)
                    with torch.cuda.stream(side_s):
                        side_s.wait_event(evs[b])
                        F_.linear_fwd(agg[r0:r1], W, bias, relu=True, ws=wsf, out=out[r0:r1])
                else:
                    F_.linear_fwd(agg[r0:r1], W, bias, relu=True, ws=wsf, out=out[r0:r1])
            if overlap:
                main_s.wait_stream(side_s)

        step()
        torch.cuda.synchronize()
        err_a = float((agg - ref_agg).abs().max())
        err_o = float((out - ref_out).abs().max())
        res[f"B{B}"] = {"bounds": bounds, "entries": [s.nnz_max for s in sides],
                        "ms_overlap": round(timeit(lambda: step(True)), 4), "ms_inline": round(timeit(lambda: step(False)), 4),
                        "ms_alt": round(timeit(step_alt), 4),
                        "agg_bit_equal": err_a == 0.0, "out_err": err_o}
        print(json.dumps({f"B{B}": res[f"B{B}"]}), flush=True)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
