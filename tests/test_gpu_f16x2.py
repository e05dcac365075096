"""The f32 projection GEMMs on TWO fp16 pieces per operand (NPI_GEMM_SPLIT_F16X2: three matrix products per tile pair instead of the
six of the bf16 x 3 split; csrc/gemm_f32.hip, EXPERIMENTS A33 / A34): accuracy against fp64 at the level of an f32 matmul, the
power-of-two row scales (npi_row_scales; written by the aggregation launch itself through npi_segsum_ex), and the layers that
use the arithmetic.  The reference's op is torch.matmul(aggr_out, self.weight) (PyG 1.4.2 SAGEConv.update; call sites
src/classes.py:62,66,70)."""
import math

import pytest
import torch

import npi_gnn_amd as npi
from npi_gnn_amd import functional as NF
from _util import GRAD_REL, rel_max
from oracle import ref_conv as R

pytestmark = pytest.mark.gpu


def _err(c, ref):
    return float((c.double() - ref).abs().max() / ref.abs().max())


@pytest.mark.parametrize("M,K,N", [(4096, 256, 256), (130, 128, 128), (1000, 512, 256), (20001, 256, 128), (777, 64, 384)])
def test_fp16x2_gemms_are_f32_accurate(dev, M, K, N):
    """forward and backward-data, prepared planes and planes made inside the call (bit-equal), bias / row scale / ReLU epilogue:
    max |C - C_fp64| / max |C| at f32-matmul level (the bf16 x 3 kernel: 6.3e-7, torch.matmul in f32: 7e-7)"""
    g = torch.Generator(device=dev).manual_seed(M)
    a = torch.randn(M, K, device=dev, generator=g)
    w = torch.randn(K, N, device=dev, generator=g) / math.sqrt(K)
    b = torch.randn(N, device=dev, generator=g)
    rs = torch.rand(M, device=dev, generator=g) + 0.5
    sc = NF.row_scales(a)
    ref_s = torch.pow(2.0, 14 - torch.floor(torch.log2(a.abs().max(dim=1).values.double()))).float()
    assert torch.equal(sc, ref_s)                                            # max |row| * scale in [2^14, 2^15)
    wsf, wsb = NF.prepare_weight(w, backward=True, f16=True)
    c = NF.linear_fwd(a, w, ws=wsf, a_scales=sc)
    assert torch.equal(c, NF.linear_fwd(a, w, a_scales=sc))
    assert _err(c, a.double() @ w.double()) <= 1e-6
    e = NF.linear_fwd(a, w, b, rowscale=rs, relu=True, ws=wsf, a_scales=sc)
    assert _err(e, torch.relu(rs[:, None].double() * (a.double() @ w.double()) + b.double())) <= 1e-6
    dc = torch.randn(M, N, device=dev, generator=g)
    d = NF.linear_bwd_data(dc, w, rowscale=rs, ws=wsb, dc_scales=NF.row_scales(dc))
    assert torch.equal(d, NF.linear_bwd_data(dc, w, rowscale=rs, dc_scales=NF.row_scales(dc)))
    assert _err(d, rs[:, None].double() * (dc.double() @ w.double().t())) <= 1e-6
    # bitwise reproducible
    assert torch.equal(c, NF.linear_fwd(a, w, ws=wsf, a_scales=sc))


def test_fp16x2_scales_rows_and_columns_of_any_magnitude(dev):
    """fp16 has five exponent bits: the row scales of A and the column scales of W (inside the preparation) put every row / column
    into range -- rows 30 decades apart, elements 8 decades apart inside a row, an all-zero row, a row of 1e-30 and one of 1e+30,
    weight columns 6 decades apart; the error is measured against every ROW's own largest |C|"""
    g = torch.Generator(device=dev).manual_seed(3)
    M, K, N = 8192, 256, 256
    a = torch.randn(M, K, device=dev, generator=g) * torch.pow(10.0, torch.rand(M, 1, device=dev, generator=g) * 30 - 15)
    a *= torch.pow(10.0, torch.rand(M, K, device=dev, generator=g) * 8 - 4)
    a[5] = 0
    a[6] *= 1e-30 / a[6].abs().max()
    a[7] *= 1e30 / a[7].abs().max()
    w = torch.randn(K, N, device=dev, generator=g) * torch.pow(10.0, torch.rand(1, N, device=dev, generator=g) * 6 - 3)
    ref = a.double() @ w.double()
    c = NF.linear_fwd(a, w, a_scales=NF.row_scales(a))
    den = ref.abs().max(dim=1, keepdim=True).values.clamp(min=1e-300)
    assert torch.isfinite(c).all() and bool((c[5] == 0).all())
    assert float(((c.double() - ref).abs() / den).max()) <= 2e-6
    # and ELEMENT by element against sum_k |a||w| -- the bound an f32 matmul itself is held to (torch.matmul: 1e-6 on these operands,
    # the bf16 x 3 kernel 7.5e-7; a column of small weights is not drowned by a large one)
    bound = (a.abs().double() @ w.abs().double()).clamp(min=1e-300)
    assert float(((c.double() - ref).abs() / bound).max()) <= 1e-6


def test_fp16x2_tiny_rows_times_tiny_columns_do_not_underflow_on_the_way_out(dev):
    """the epilogue undoes the two power-of-two scales ONE AFTER THE OTHER, (acc 2^-eb) 2^-ea -- folded into one
    factor their product would underflow when both a row of A and a column of W are tiny although the result is representable.
    Rows of A at 2^-60 against columns of W at 2^-60: products at 2^-120, normal f32 numbers, within the kernel's bars; and
    against the bf16 x 3 arithmetic on the same operands."""
    g = torch.Generator(device=dev).manual_seed(12)
    M, K, N = 1024, 256, 256
    a = torch.randn(M, K, device=dev, generator=g)
    w = torch.randn(K, N, device=dev, generator=g) / K ** 0.5
    a[::2] *= 2.0 ** -60                                                     # every other row tiny
    w[:, ::3] *= 2.0 ** -60                                                  # every third column tiny
    ref = a.double() @ w.double()
    c2 = NF.linear_fwd(a, w, a_scales=NF.row_scales(a))
    c3 = NF.linear_fwd(a, w)
    assert float(ref[::2, ::3].abs().max()) < 2.0 ** -100 and float(ref[::2, ::3].abs().min()) > 2.0 ** -140    # tiny, normal
    bound = (a.abs().double() @ w.abs().double())
    for c in (c2, c3):
        assert torch.isfinite(c).all() and bool((c[::2, ::3] != 0).all())
        assert float(((c.double() - ref).abs() / bound).max()) <= 1e-6


def test_weight_gradient_on_fp16x2_with_column_scales(dev):
    """dW = A^T dC on two fp16 pieces per operand (linear_bwd_weight(a_cs=, dc_cs=)): the contraction runs over the rows, so the
    power-of-two scales that factor out are per COLUMN (npi_col_scales).  Against fp64 at the bar of the bf16 x 3 kernel -- max
    |diff| / max |ref| <= 1e-5 -- with true column scales on columns 12 decades apart (then also per dW ROW), and with the uniform
    scale derived from the operands' ROW scales (no pass over either matrix: what GATConv's backward does); db unchanged."""
    g = torch.Generator(device=dev).manual_seed(21)
    M, K, N = 70_001, 256, 256                                              # (a ragged tail of rows: the finishing kernel's share)
    a = torch.randn(M, K, device=dev, generator=g)
    dc = torch.randn(M, N, device=dev, generator=g)
    ref = a.double().t() @ dc.double()
    dw3, db3 = NF.linear_bwd_weight(a, dc)
    dw2, db2 = NF.linear_bwd_weight(a, dc, a_cs=NF.col_scales(a), dc_cs=NF.col_scales(dc))
    assert not torch.equal(dw2, dw3) and torch.equal(db2, db3)
    assert rel_max(dw2, ref) <= GRAD_REL and rel_max(dw3, ref) <= GRAD_REL
    # the uniform scale from the row scales
    dwu, _ = NF.linear_bwd_weight(a, dc, a_cs=NF.col_scales(row_scales=NF.row_scales(a), cols=K),
                                  dc_cs=NF.col_scales(row_scales=NF.row_scales(dc), cols=N))
    assert rel_max(dwu, ref) <= GRAD_REL
    sc = NF.col_scales(row_scales=NF.row_scales(a), cols=K)
    assert bool((sc == sc[0]).all()) and float(sc[0]) == float(NF.row_scales(a).min())
    # columns 12 decades apart: per-column scales keep every ROW of dW (a column of A) at its own relative accuracy
    a2 = a * torch.pow(10.0, torch.linspace(-6, 6, K, device=dev)).view(1, K)
    ref2 = a2.double().t() @ dc.double()
    dwc, _ = NF.linear_bwd_weight(a2, dc, a_cs=NF.col_scales(a2), dc_cs=NF.col_scales(dc))
    row_err = (dwc.double() - ref2).abs().max(dim=1).values / ref2.abs().max(dim=1).values
    assert float(row_err.max()) <= GRAD_REL
    cs = NF.col_scales(a2)
    for k in (0, 100, 255):
        assert float(cs[k]) == float(NF.row_scales(a2[:, k].contiguous().view(1, -1))[0])   # the column's scale = its own row scale


def test_fp16x2_non_finite_operands_behave_as_under_the_bf16_split(dev):
    """NaN stays NaN; an Inf in A makes its output row NaN (Inf - Inf in the split), as the bf16 x 3 arithmetic does (include/npi_gnn.h)"""
    g = torch.Generator(device=dev).manual_seed(4)
    a = torch.randn(256, 128, device=dev, generator=g)
    w = torch.randn(128, 128, device=dev, generator=g)
    a[3, 5] = float("nan")
    a[9, 7] = float("inf")
    c = NF.linear_fwd(a, w, a_scales=NF.row_scales(a))
    assert torch.isnan(c[3]).all() and torch.isnan(c[9]).all()
    keep = torch.ones(256, dtype=torch.bool, device=dev)
    keep[3] = keep[9] = False
    assert torch.isfinite(c[keep]).all() and _err(c[keep], a[keep].double() @ w.double()) <= 1e-6


def test_aggregation_writes_the_row_scales_its_projection_needs(dev):
    """npi_segsum_ex: the finished rows' power-of-two scales from the aggregation launch itself -- bit-equal to a pass over the
    finished matrix (npi_row_scales), for mean / weighted sums, hub rows cut across items, empty rows, both item sizes"""
    N, E = 30_000, 400_000
    g = torch.Generator().manual_seed(5)
    ei = torch.randint(0, N, (2, E), generator=g)
    ei[1, : E // 3] = 7                                                      # a hub row: cut across many items and workgroups
    ei = ei[:, ei[1] != 11]                                                  # row 11: only its self loop
    x = torch.randn(N, 256, generator=g).to(dev)
    for item in (64, 256):
        graph = npi.CSRGraph(ei.to(dev), N, item=item)
        for w, mean in ((None, True), (torch.rand(graph.by_dst.nnz_max, device=dev), False)):
            plain = NF.segsum(graph, graph.by_dst, x, w=w, mean=mean)
            sc = torch.empty(N, device=dev)
            out = NF.segsum(graph, graph.by_dst, x, w=w, mean=mean, scales_out=sc)
            assert torch.equal(out, plain)                                   # the 7-wave variant sums in the same order
            assert torch.equal(sc, NF.row_scales(out))
    assert not NF.segsum_scales_ok(graph.by_dst, x[:, :128].contiguous())   # 256 columns only


@pytest.mark.parametrize("kind", ["sage", "gcn"])
def test_layers_on_fp16x2_projections_match_the_oracle(dev, kind):
    """Schedule(f16x2_min_rows=0): both projections of SAGEConv / GCNConv (aggregate first) at 256 features on the fp16 x 2 kernel --
    the forward's row scales from the aggregation launch, the backward run aggregate-first (dX = (A^T dOut) W^T) with the scales
    from the transposed aggregation -- outputs and every gradient at the layers' own bars, and within rounding of the bf16 x 3 run;
    Schedule(aggregate_first_backward=False): the backward in PyG's literal order on bf16 x 3, the same numbers"""
    N, E, F = 3000, 30_000, 256
    g = torch.Generator().manual_seed(6)
    ei = torch.randint(0, N, (2, E), generator=g)
    x = torch.randn(N, F, generator=g)
    go = torch.randn(N, F, generator=g)
    conv = (npi.SAGEConv(F, F) if kind == "sage" else npi.GCNConv(F, F)).to(dev)
    with torch.no_grad():
        conv.bias.copy_(torch.randn(F, generator=g) * 0.1)
    graph = npi.CSRGraph(ei.to(dev), N)
    from npi_gnn_amd.schedule import DEFAULT
    res = {}
    for rows, sch in ((None, DEFAULT.but(f16x2_min_rows=None)), (0, DEFAULT.but(f16x2_min_rows=0)),
                      ("literal", DEFAULT.but(f16x2_min_rows=0, aggregate_first_backward=False))):
        conv.schedule = sch
        conv.zero_grad()
        xd = x.to(dev).requires_grad_(True)
        out = conv(xd, graph)
        out.backward(go.to(dev))
        res[rows] = (out.detach(), xd.grad, conv.weight.grad.clone(), conv.bias.grad.clone())
    assert not torch.equal(res[0][0], res[None][0]) and rel_max(res[0][0], res[None][0]) <= 2e-6     # another arithmetic, the same numbers
    assert torch.equal(res["literal"][0], res[0][0])                          # the forward does not depend on the backward's order
    assert not torch.equal(res["literal"][1], res[0][1]) and rel_max(res["literal"][1], res[0][1]) <= 5e-6
    assert rel_max(res[None][1], res[0][1]) <= 5e-6
    for k in (2, 3):                                                         # dW / db: the same launch on the same operands either way
        assert torch.equal(res["literal"][k], res[0][k])
    x6, W6, b6 = (t.detach().cpu().double().clone().requires_grad_(True) for t in (x, conv.weight, conv.bias))
    ref = (R.sage_conv if kind == "sage" else R.gcn_conv)(x6, ei, W6, b6)
    ref.backward(go.double())
    out, dx, dW, db = res[0]
    assert float((out.cpu().double() - ref.detach()).abs().max()) <= 1e-4 * max(1.0, float(ref.abs().max()))
    assert float((dx.cpu().double() - x6.grad).abs().max()) <= 1e-4 * max(1.0, float(x6.grad.abs().max()))
    assert rel_max(dW, W6.grad) <= GRAD_REL and rel_max(db, b6.grad) <= GRAD_REL


def test_gat_backward_projects_on_fp16x2_with_scales_from_the_fused_pass(dev):
    """one-head GATConv, 256 channels: dX = d hfeat W^T (+ the rank-2 attention terms in the store epilogue) on the fp16 x 2 kernel,
    the row scales of d hfeat written by the fused by-source pass (npi_gat_backward_fused_heads) -- bit-equal to a pass over
    d hfeat; every gradient at the layer's bars against the oracle and within rounding of the bf16 x 3 run"""
    from npi_gnn_amd.schedule import DEFAULT
    N, E, Fi, C = 3000, 40_000, 128, 256
    g = torch.Generator().manual_seed(8)
    ei = torch.randint(0, N, (2, E), generator=g)
    ei[0, : E // 4] = 5                                                      # a heavy SOURCE row: cut across items in the by-source pass
    x = torch.randn(N, Fi, generator=g)
    W = (torch.rand(Fi, C, generator=g) * 2 - 1) * (6.0 / (Fi + C)) ** 0.5
    att = (torch.rand(1, 1, 2 * C, generator=g) * 2 - 1) * 0.3
    b = torch.randn(C, generator=g) * 0.1
    go = torch.randn(N, C, generator=g)
    graph = npi.CSRGraph(ei.to(dev), N)
    res = {}
    for rows in (None, 0):
        sch = DEFAULT.but(gat_rank2_min_rows=0, f16x2_min_rows=rows)
        xd, Wd, ad, bd = (t.to(dev).requires_grad_(True) for t in (x, W, att, b))
        out = npi.gat_conv(xd, graph, Wd, ad, bd, heads=1, schedule=sch)
        out.backward(go.to(dev))
        res[rows] = (out.detach(), xd.grad, Wd.grad, ad.grad, bd.grad)
    assert torch.equal(res[0][0], res[None][0])                              # the forward is untouched
    assert not torch.equal(res[0][1], res[None][1]) and rel_max(res[0][1], res[None][1]) <= 2e-6
    for k in (2, 3, 4):
        assert torch.equal(res[0][k], res[None][k])                          # dW / d att / db do not pass through that GEMM
    xr, Wr, ar, br = (t.clone().double().requires_grad_(True) for t in (x, W, att, b))
    R.gat_conv(xr, ei, Wr, ar, br, heads=1).backward(go.double())
    assert torch.allclose(res[0][1].cpu(), xr.grad.float(), atol=2e-4, rtol=1e-3)
    # the scales the fused pass writes are those of its output
    sr = graph.by_src
    hfeat = (x @ W).to(dev)
    a_dst, a_src = NF.gat_scores(hfeat, att.view(1, 2 * C).to(dev), 1, C)
    m, s = NF.gat_softmax_stats(graph.by_dst, a_dst, a_src, 1, 0.2)
    D = NF.gat_rowdot(go.to(dev), torch.zeros(N, C, device=dev), None, 1, C)
    tpack = NF.gat_pack_targets(a_dst, m, s, D)
    sc = torch.empty(N, device=dev)
    dh, _ = NF.gat_backward_fused_packed(sr, go.to(dev), None, hfeat, C, tpack, a_src, 0.2, scales_out=sc)
    dh0, _ = NF.gat_backward_fused_packed(sr, go.to(dev), None, hfeat, C, tpack, a_src, 0.2)
    assert torch.equal(dh, dh0) and torch.equal(sc, NF.row_scales(dh))


def test_gat_stack_hands_row_scales_from_layer_to_layer(dev):
    """gat_conv(x_scales=, return_scales=True): the scales of the input features computed once, every layer's projection x W on the
    fp16 x 2 kernel (row-dot epilogue), the scales of a layer's output -- bias and ReLU applied -- written by its aggregation launch
    and bit-equal to a pass over the output; outputs and gradients of the 2-layer stack within rounding of the plain run and at
    the layers' bars against the oracle"""
    N, E, F = 4000, 50_000, 256
    g = torch.Generator().manual_seed(9)
    ei = torch.randint(0, N, (2, E), generator=g)
    ei[1, : E // 4] = 3
    x = torch.randn(N, F, generator=g)
    Ws = [(torch.rand(F, F, generator=g) * 2 - 1) * (6.0 / (2 * F)) ** 0.5 for _ in range(2)]
    atts = [(torch.rand(1, 1, 2 * F, generator=g) * 2 - 1) * 0.3 for _ in range(2)]
    bs = [torch.randn(F, generator=g) * 0.1 for _ in range(2)]
    go = torch.randn(N, F, generator=g)
    graph = npi.CSRGraph(ei.to(dev), N)
    from npi_gnn_amd.schedule import DEFAULT
    res = {}
    for mode in ("plain", "scales"):
        sch = DEFAULT.but(f16x2_min_rows=None if mode == "plain" else 0)
        P = [t.to(dev).requires_grad_(True) for t in Ws + atts + bs]
        xd = x.to(dev).requires_grad_(True)
        h, hs = xd, (NF.row_scales(xd.detach()) if mode == "scales" else None)
        for k in range(2):
            h, hs = npi.gat_conv(h, graph, P[k], P[2 + k], P[4 + k], heads=1, relu=True, x_scales=hs, return_scales=True, schedule=sch)
            if mode == "scales":
                assert hs is not None and torch.equal(hs, NF.row_scales(h.detach()))
        h.backward(go.to(dev))
        res[mode] = [h.detach(), xd.grad] + [p.grad for p in P]
    for a, b in zip(res["scales"], res["plain"]):
        assert rel_max(a, b) <= 5e-6
    assert not torch.equal(res["scales"][0], res["plain"][0])
    P6 = [t.clone().double().requires_grad_(True) for t in Ws + atts + bs]
    x6 = x.clone().double().requires_grad_(True)
    h = x6
    for k in range(2):
        h = torch.relu(R.gat_conv(h, ei, P6[k], P6[2 + k], P6[4 + k], heads=1))
    h.backward(go.double())
    out, dx = res["scales"][0], res["scales"][1]
    assert float((out.cpu().double() - h.detach()).abs().max()) <= 1e-4 * max(1.0, float(h.abs().max()))
    assert float((dx.cpu().double() - x6.grad).abs().max()) <= 2e-4 * max(1.0, float(x6.grad.abs().max()))
    for got, want in zip(res["scales"][2:], P6):
        assert rel_max(got, want.grad) <= 3 * GRAD_REL


def test_row_scales_of_the_wrong_shape_are_refused_before_any_launch(dev):
    """one scale per row of the left operand: a shorter vector would be an out-of-bounds device read"""
    a = torch.randn(512, 128, device=dev)
    w = torch.randn(128, 128, device=dev)
    for bad in (torch.ones(511, device=dev), torch.ones(512, 1, device=dev), torch.ones(512, device=dev, dtype=torch.float64),
                torch.ones(1024, device=dev)[::2]):
        with pytest.raises(ValueError):
            NF.linear_fwd(a, w, a_scales=bad)
        with pytest.raises(ValueError):
            NF.linear_bwd_data(a, w.t().contiguous(), dc_scales=bad)
    conv = npi.GATConv(128, 128).to(dev)
    ei = torch.randint(0, 512, (2, 4000), device=dev)
    with pytest.raises(ValueError):
        conv(a, ei, x_scales=torch.ones(100, device=dev))
