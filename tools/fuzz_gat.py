#!/usr/bin/env python3
"""Randomised campaign for the GAT kernels against the CPU oracle (oracle/ref_conv.py): random graphs (hub rows, empty
rows, self loops, duplicate edges), 1 / 2 / 4 / 8 heads, both item sizes (CSRGraph(item=)), rows in list or in column order
(CSRGraph(sort_columns=)), fused ReLU on / off, attention dropout in training mode (a quarter of the cases; the same draw handed to
the oracle).
usage: tools/fuzz_gat.py [cases] [seed [only_case [sort_columns 0|1]]]"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import npi_gnn_amd as npi
from npi_gnn_amd.schedule import DEFAULT
from oracle import ref_conv as R
# small graphs too: the rank-2 store epilogue of dX and the second-stream schedule of the backward (the product takes both from
# 100,000 rows on); every other case keeps the one-stream arrangement
SCHS = [DEFAULT.but(gat_rank2_min_rows=0), DEFAULT.but(gat_rank2_min_rows=0, overlap_min_rows=0)]

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
only = int(sys.argv[3]) if len(sys.argv) > 3 else None          # re-run ONE case of the campaign ...
force_sort = {"0": False, "1": True}.get(sys.argv[4]) if len(sys.argv) > 4 else None     # ... with the row order forced
force_item = int(sys.argv[5]) if len(sys.argv) > 5 else None                            # ... and the item size
dev = torch.device("cuda:0")
worst = 0.0
flips = 0          # cases set aside: the fp32 ORACLE is as far from the fp64 one as the kernels are
for it in range(cases):
    H, C = [(1, 256), (1, 64), (1, 100), (2, 32), (2, 128), (4, 64), (4, 32), (8, 32), (2, 64), (1, 8)][int(rng.integers(0, 10))]
    N = int(rng.integers(2, 2500))
    E = int(rng.integers(0, 40000))
    Fi = int(rng.choice([16, 33, 64, 128, 128, 256]))       # (128 / 256 with one head: the rank-2 store epilogue of dX)
    big_items = bool(rng.random() < 0.5)
    g = torch.Generator().manual_seed(int(rng.integers(0, 2 ** 31)))
    ei = torch.randint(0, N, (2, E), generator=g)
    if E and rng.random() < 0.5:
        ei[1, : E // 2] = int(rng.integers(0, N))
    if E and rng.random() < 0.3:
        ei[0, E // 2:] = int(rng.integers(0, N))
    relu = bool(rng.random() < 0.3)
    x = torch.randn(N, Fi, generator=g)
    W = (torch.rand(Fi, H * C, generator=g) * 2 - 1) * (6.0 / (Fi + H * C)) ** 0.5
    att = (torch.rand(1, H, 2 * C, generator=g) * 2 - 1) * (6.0 / (H + 2 * C)) ** 0.5 * 3.0
    b = torch.randn(H * C, generator=g) * 0.1
    go = torch.randn(N, H * C, generator=g)
    xd, Wd, ad, bd = (t.to(dev).requires_grad_(True) for t in (x, W, att, b))
    sort_columns = bool(rng.random() < 0.5)
    if only is not None and it != only:
        continue
    if force_sort is not None:
        sort_columns = force_sort
    if force_item is not None:
        big_items = force_item == 256
    graph = npi.CSRGraph(ei.to(dev), N, item=256 if big_items else 64, sort_columns=sort_columns)
    # every other pair of cases: dX's GEMM on two fp16 pieces per operand wherever the shape takes it (one head of 256 channels, the
    # rank-2 epilogue), the row scales from the fused backward pass -- the product switches it on from 100,000 rows
    f16_rows = {"off": 100_000, "on": 0}.get(os.environ.get("FUZZ_F16", "toggle"), 0 if (it & 2) else 100_000)
    drop = bool(rng.random() < 0.25)
    keep = ks = None
    if drop:
        from npi_gnn_amd.functional import gat_dropout_keep
        keep = gat_dropout_keep(graph, H, float(rng.choice([0.1, 0.5, 0.8])))
        nnz = int(graph.by_dst.rowptr[-1])
        ks = R.keep_scale_from_entries(ei, N, graph.by_dst.eid[:nnz].cpu(), graph.by_dst.rowidx[:nnz].cpu(), keep[:nnz].cpu())
    out = npi.gat_conv(xd, graph, Wd, ad, bd, heads=H, relu=relu, schedule=SCHS[it & 1].but(f16x2_min_rows=f16_rows), keep=keep)
    out.backward(go.to(dev))
    xr, Wr, ar, br = (t.clone().double().requires_grad_(True) for t in (x, W, att, b))
    ref = R.gat_conv(xr, ei, Wr, ar, br, heads=H, keep_scale=ks)
    if relu:
        # the ReLU mask of the run under test: a pre-activation within rounding of zero may have either sign in the fp64
        # oracle and in the fp32 kernels, and ONE flipped element moves db by |dOut| (seed 2, case 71); outputs are still
        # compared against the oracle's own ReLU below, up to that rounding
        ref = ref * (out.detach().cpu() > 0).double()
    ref.backward(go.double())
    errs = []
    for name, a, r in (("out", out.detach(), ref.detach()), ("dx", xd.grad, xr.grad), ("dW", Wd.grad, Wr.grad),
                       ("datt", ad.grad, ar.grad), ("db", bd.grad, br.grad)):
        # d att sums g h over ALL nodes, and g (a row sum of dz) cancels to ~0 when leaky_relu' is constant over a row: on a
        # degenerate multigraph (2 nodes, 38 k parallel edges: seed 83, case 11) the reference itself is ~1e-3 and f32
        # cancellation noise of 2e-6 absolute reads as 2e-3 "relative" -- the floor of the scale is 5e-2 for that tensor (the other gradients of the case are O(1))
        scale = max(float(r.abs().max()), 5e-2 if name == "datt" else 1e-3)
        e = float((a.cpu().double() - r).abs().max()) / scale
        errs.append((name, e, float(r.abs().max())))
    m = max(e for _, e, _ in errs)
    if m <= 2e-4:
        worst = max(worst, m)
    if m > 2e-4:
        if os.environ.get("FUZZ_DUMP"):                       # the inputs of the first mismatch, for a replay outside the campaign
            torch.save({"ei": ei, "x": x, "W": W, "att": att, "b": b, "go": go, "H": H, "C": C, "N": N, "relu": relu, "big_items": big_items,
                        "sort_columns": sort_columns, "keep": None if keep is None else keep.cpu()}, os.environ["FUZZ_DUMP"])
        print(f"MISMATCH case {it}: sort_columns={sort_columns} H={H} C={C} N={N} E={E} Fi={Fi} big_items={big_items} relu={relu} drop={drop}: {errs}")
        # is it the data?  the ORACLE evaluated in fp32 against itself in fp64: what rounding alone does to this case
        x32, W32, a32, b32 = (t.clone().float().requires_grad_(True) for t in (x, W, att, b))
        r32 = R.gat_conv(x32, ei, W32, a32, b32, heads=H, keep_scale=None if ks is None else ks.float())
        if relu:
            r32 = r32 * (out.detach().cpu() > 0).float()
        r32.backward(go.float())
        floor = {n: float((a_.double() - r_).abs().max()) / max(float(r_.abs().max()), 1e-3) for n, a_, r_ in (
            ("dx", x32.grad, xr.grad), ("dW", W32.grad, Wr.grad), ("datt", a32.grad, ar.grad))}
        print(f"   the oracle in fp32 against the oracle in fp64: {floor}")
        if all(e <= 2.0 * floor.get(n, 0.0) + 2e-4 for n, e, _ in errs):
            # a leaky_relu argument within rounding of zero has either sign in fp32 and in fp64, and its slope jumps from 1 to
            # 0.2: EVERY fp32 evaluation is that far from the fp64 one (seed 23, case 107: the oracle in fp32 is off by the same
            # 5.4e-3 to six digits) -- the data, not the kernels
            print("   -> the fp32 oracle is as far from the fp64 one: a discontinuity of the data (leaky_relu at a rounding-level "
                  "argument), not counted")
            flips += 1
            continue
        # the same discontinuity when the fp32 ORACLE happens to land on the fp64 side of zero and the kernels -- which form the score
        # as a_dst[i] + a_src[j] from two row dots, another rounding -- on the other (seed 51, case 104: z = 5.6e-7 from two
        # halves of magnitude 4; the one entry's dz comes out 0.2 x, the forward agrees to 7e-7): look for it directly
        with torch.no_grad():
            e2 = R.add_self_loops(R.remove_self_loops(ei), N)
            hh = (x.double() @ W.double()).view(N, H, C)
            a_i = (hh * att.double()[0, :, :C]).sum(-1)[e2[1]]
            a_j = (hh * att.double()[0, :, C:]).sum(-1)[e2[0]]
            near = float(((a_i + a_j).abs() / (a_i.abs() + a_j.abs()).clamp(min=1e-30)).min())
        if near < 1e-6 and errs[0][1] <= 2e-5:
            print(f"   -> a leaky_relu argument is {near:.1e} of its two halves: its sign is decided by rounding (the forward agrees to "
                  f"{errs[0][1]:.1e}); a discontinuity of the data, not counted")
            flips += 1
            continue
        # bisect: the same inputs with single arrangements switched off
        SCH = SCHS[it & 1]
        for name, sch in (("as run", SCH), ("no rank-2 epilogue", SCH.but(gat_rank2_epilogue=False)),
                          ("no scores epilogue", SCH.but(gat_scores_epilogue=False)), ("one stream", SCH.but(overlap_streams=False))):
            for use_relu in ((relu, False) if relu else (False,)):
                xd, Wd, ad, bd = (t.to(dev).requires_grad_(True) for t in (x, W, att, b))
                o = npi.gat_conv(xd, graph, Wd, ad, bd, heads=H, relu=use_relu, schedule=sch, keep=keep)
                if relu and not use_relu:
                    o = torch.relu(o)
                o.backward(go.to(dev))
                e = float((xd.grad.cpu().double() - xr.grad).abs().max()) / max(float(xr.grad.abs().max()), 1e-3)
                print(f"   {name}, fused_relu={use_relu}: dx err {e:.2e}")
        sys.exit(1)
print(f"{cases} cases ok, worst relative error {worst:.2e} ({flips} case(s) set aside: fp32 oracle equally far from fp64)")
