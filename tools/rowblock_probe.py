#!/usr/bin/env python3
"""Forward of one SAGEConv layer at C4 with the aggregation cut into two ROW BLOCKS (two CSR sides), so that block 1's
projection GEMM (on fewer CUs: NPI_GEMM_RESERVE_CUS) runs beside block 2's gathers: agg1 -> (GEMM1 || agg2) -> GEMM2, against
agg -> GEMM.  usage: tools/rowblock_probe.py [nodes edges]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import npi_gnn_amd as npi
from npi_gnn_amd import functional as NF
from npi_gnn_amd.graph import build_side
from npi_gnn_amd.synth import bipartite_edge_index
N, E = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1_000_000, 20_000_000)
dev, F = torch.device("cuda:0"), 256
g = torch.Generator().manual_seed(1)
ei = bipartite_edge_index(N, E, seed=20260310).to(dev)
loops = torch.arange(N, device=dev)
src, dst = torch.cat([ei[0], loops]), torch.cat([ei[1], loops])
x = torch.randn(N, F, generator=g).to(dev)
W = (torch.randn(F, F, generator=g) / 16).to(dev)
b = torch.randn(F, generator=g).to(dev)
full = build_side(dst, src, N, N, False, 0, False)
wsf, _ = NF.prepare_weight(W, backward=False)
side_stream = torch.cuda.Stream(device=dev)


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def plain():
    agg = NF.segsum(None, full, x, mean=True)
    return NF.linear_fwd(agg, W, b, ws=wsf), agg


ref, ref_agg = plain()
print(f"agg -> GEMM: {timeit(plain):.3f} ms   (aggregation alone {timeit(lambda: NF.segsum(None, full, x, mean=True)):.3f}, "
      f"GEMM alone {timeit(lambda: NF.linear_fwd(ref_agg, W, b, ws=wsf)):.3f})")
for frac in (0.5, 0.7, 0.8, 0.9):
    R = int(N * frac) // 128 * 128
    m = dst < R
    s1 = build_side(dst[m], src[m], R, N, False, 0, False)
    s2 = build_side(dst[~m] - R, src[~m], N - R, N, False, 0, False)
    for rc in (0, 64, 128):
        agg = torch.empty(N, F, device=dev)
        out = torch.empty(N, F, device=dev)

        def blocks():
            main = torch.cuda.current_stream(dev)
            NF.segsum(None, s1, x, mean=True, out=agg[:R])
            ev = torch.cuda.Event(); ev.record(main)
            NF.linear_fwd(agg[:R], W, b, out=out[:R], ws=wsf, reserve_cus=rc)
            with torch.cuda.stream(side_stream):
                side_stream.wait_event(ev)
                NF.segsum(None, s2, x, mean=True, out=agg[R:])
            main.wait_stream(side_stream)
            NF.linear_fwd(agg[R:], W, b, out=out[R:], ws=wsf)
            return out
        t = timeit(blocks)
        same = torch.equal(blocks(), ref)
        print(f"rows [0, {frac:.1f} N) first, GEMM1 leaves {rc:3d} CUs: {t:.3f} ms   bit-equal to the plain forward: {same}")
