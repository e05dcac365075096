"""GATConv on the HIP path vs the CPU oracle (PyG 1.4.2 formulas restated; parity unpinned by any
reference artifact -- SURVEY.md 8(c)).  Gradients of the oracle come from torch autograd through the
restated ops (themselves gradcheck'ed in fp64 in test_oracle_cpu.py-style below)."""
import pytest
import torch

import npi_gnn_amd as npi
from _util import GRAD_REL, rel_max
from oracle import ref_conv as R

pytestmark = pytest.mark.gpu


def _case(N, E, Fi, H, C, seed, hub=True):
    g = torch.Generator().manual_seed(seed)
    ei = torch.randint(0, N, (2, E), generator=g)
    if hub:
        ei[1, : E // 3] = 3                    # one target with ~E/3 entries: workgroup-per-row softmax path
        ei[0, E // 3: E // 2] = 5              # one heavy source row for the backward
    x = torch.randn(N, Fi, generator=g)
    W = (torch.rand(Fi, H * C, generator=g) * 2 - 1) * (6.0 / (Fi + H * C)) ** 0.5
    att = (torch.rand(1, H, 2 * C, generator=g) * 2 - 1) * (6.0 / (H + 2 * C)) ** 0.5 * 3.0
    b = torch.randn(H * C, generator=g) * 0.1
    go = torch.randn(N, H * C, generator=g)
    return ei, x, W, att, b, go


@pytest.mark.parametrize("N,E,Fi,H,C", [(60, 300, 16, 1, 8), (3000, 40000, 128, 1, 256), (2000, 20000, 64, 4, 64),
                                        (1500, 12000, 178, 2, 32),
                                        (20000, 1_060_000, 32, 1, 32)])           # >= 2^20 entries: 256-entry items
def test_gat_conv_fwd_bwd_matches_oracle(dev, N, E, Fi, H, C):
    ei, x, W, att, b, go = _case(N, E, Fi, H, C, seed=N)
    # the oracle runs in fp64 when a hub row has > 10^5 entries: in fp32 ITS sum over the row is off by 3e-3
    # (the kernel stays within 3e-6 of fp64 there)
    dt = torch.float64 if E > 500_000 else torch.float32
    xr, Wr, ar, br = (t.clone().to(dt).requires_grad_(True) for t in (x, W, att, b))
    ref = R.gat_conv(xr, ei, Wr, ar, br, heads=H)
    ref.backward(go.to(dt))
    xd, Wd, ad, bd = (t.to(dev).requires_grad_(True) for t in (x, W, att, b))
    out = npi.gat_conv(xd, ei.to(dev), Wd, ad, bd, heads=H)
    out.backward(go.to(dev))
    assert torch.allclose(out.detach().cpu(), ref.detach().float(), atol=1e-4, rtol=1e-4)
    assert torch.allclose(xd.grad.cpu(), xr.grad.float(), atol=2e-4, rtol=1e-3)
    # parameter gradients: max |diff| / max |ref| <= 1e-5 against the oracle in fp64
    if dt != torch.float64:
        xr, Wr, ar, br = (t.clone().double().requires_grad_(True) for t in (x, W, att, b))
        R.gat_conv(xr, ei, Wr, ar, br, heads=H).backward(go.double())
    for got, want in ((Wd.grad, Wr.grad), (ad.grad, ar.grad), (bd.grad, br.grad)):
        assert rel_max(got, want) <= GRAD_REL, rel_max(got, want)


@pytest.mark.parametrize("N,E,Fi,H,C,concat", [(60, 300, 16, 1, 8, True), (1500, 12000, 64, 2, 32, True), (800, 6000, 32, 3, 16, False),
                                               (3000, 40000, 128, 1, 256, True)])
def test_gat_conv_with_attention_dropout_matches_oracle_under_the_same_mask(dev, N, E, Fi, H, C, concat):
    """GATConv(dropout > 0) in TRAINING mode (PyG 1.4.2 GATConv.message: alpha = F.dropout(softmax(alpha), p, training)): the mask
    drawn on the device, in by-target entry order, is carried over to the oracle's edge order (original columns without self
    loops, then the N appended loops) through the CSR's own eid / rowidx arrays; forward and every gradient at the layer's bars."""
    from npi_gnn_amd import functional as NF
    ei, x, W, att, b, go = _case(N, E, Fi, H, C, seed=N + 7)
    if not concat:
        b, go = b[:C].clone(), go[:, :C].clone()
    graph = npi.CSRGraph(ei.to(dev), N)
    keep = NF.gat_dropout_keep(graph, H, 0.4)
    d = graph.by_dst
    nnz = int(d.rowptr[-1])
    ks = R.keep_scale_from_entries(ei, N, d.eid[:nnz].cpu(), d.rowidx[:nnz].cpu(), keep[:nnz].cpu())
    frac = float((keep[:nnz] > 0).float().mean())
    assert abs(frac - 0.6) < (0.15 if nnz < 1000 else 0.03) and set(keep[:nnz].unique().tolist()) <= {0.0, float(torch.tensor(1 / 0.6))}
    xr, Wr, ar, br = (t.clone().double().requires_grad_(True) for t in (x, W, att, b))
    ref = R.gat_conv(xr, ei, Wr, ar, br, heads=H, concat=concat, keep_scale=ks)
    ref.backward(go.double())
    xd, Wd, ad, bd = (t.to(dev).requires_grad_(True) for t in (x, W, att, b))
    out = npi.gat_conv(xd, graph, Wd, ad, bd, heads=H, concat=concat, keep=keep)
    out.backward(go.to(dev))
    assert torch.allclose(out.detach().cpu(), ref.detach().float(), atol=1e-4, rtol=1e-4)
    assert torch.allclose(xd.grad.cpu(), xr.grad.float(), atol=2e-4, rtol=1e-3)
    for got, want in ((Wd.grad, Wr.grad), (ad.grad, ar.grad), (bd.grad, br.grad)):
        assert rel_max(got, want) <= GRAD_REL, rel_max(got, want)
    # the mask matters (the test would not notice a path that ignored it otherwise) and keep = 1 everywhere is the plain layer
    plain = npi.gat_conv(xd.detach(), graph, Wd.detach(), ad.detach(), bd.detach(), heads=H, concat=concat)
    assert not torch.allclose(out.detach(), plain, atol=1e-3)
    ones = npi.gat_conv(xd.detach(), graph, Wd.detach(), ad.detach(), bd.detach(), heads=H, concat=concat, keep=torch.ones_like(keep))
    assert torch.allclose(ones, plain, atol=1e-5, rtol=1e-5)


def test_gat_heavy_row_softmax(dev):
    """> 4096 entries on one target row: the workgroup-per-row statistics kernel."""
    N, E, Fi, H, C = 9000, 30000, 32, 1, 64
    ei, x, W, att, b, go = _case(N, E, Fi, H, C, seed=1)
    ref = R.gat_conv(x, ei, W, att, b, heads=H)
    out = npi.gat_conv(x.to(dev), ei.to(dev), W.to(dev), att.to(dev), b.to(dev), heads=H)
    assert torch.allclose(out.cpu(), ref, atol=1e-4, rtol=1e-4)


def test_gat_module_concat_false_and_state_dict(dev):
    torch.manual_seed(0)
    N, E = 500, 3000
    ei = torch.randint(0, N, (2, E))
    x = torch.randn(N, 32)
    conv = npi.GATConv(32, 16, heads=3, concat=False)
    assert [tuple(v.shape) for v in conv.state_dict().values()] == [(32, 48), (1, 3, 32), (16,)]
    ref = R.gat_conv(x, ei, conv.weight.detach(), conv.att.detach(), conv.bias.detach(), heads=3, concat=False)
    conv = conv.to(dev)
    out = conv(x.to(dev), ei.to(dev))
    assert torch.allclose(out.detach().cpu(), ref, atol=1e-4, rtol=1e-4)
    out.sum().backward()
    assert conv.att.grad is not None and torch.isfinite(conv.att.grad).all()


@pytest.mark.parametrize("H,C", [(1, 256), (4, 64), (8, 32)])
def test_gat_is_bitwise_reproducible(dev, H, C):
    """forward AND every gradient, run to run: no float atomics anywhere (item scans, fused backward, chain folds)"""
    ei, x, W, att, b, go = _case(4000, 60000, 64, H, C, seed=4)
    runs = []
    for _ in range(2):
        xd, Wd, ad, bd = (t.to(dev).requires_grad_(True) for t in (x, W, att, b))
        out = npi.gat_conv(xd, ei.to(dev), Wd, ad, bd, heads=H)
        out.backward(go.to(dev))
        runs.append((out.detach(), xd.grad, Wd.grad, ad.grad, bd.grad))
    for a, c in zip(*runs):
        assert torch.equal(a, c)


def _sides(dev, n_rows, n_cols, E, seed, hub_rows=(3,), empty_from=None):
    """a by-row CSR side over random entries (no self loops added): some rows very long, the tail rows empty"""
    from npi_gnn_amd import graph as NG
    g = torch.Generator().manual_seed(seed)
    hi = empty_from if empty_from is not None else n_rows
    key = torch.randint(0, hi, (E,), generator=g)
    for k, r in enumerate(hub_rows):
        key[k * (E // 4): k * (E // 4) + E // 5] = r
    val = torch.randint(0, n_cols, (E,), generator=g)
    side = NG.build_side(key.to(dev), val.to(dev), n_rows, n_cols, False, 0, False)
    return side, key, val


@pytest.mark.parametrize("n_rows,E,H", [(50, 300, 1), (5000, 90_000, 1), (3000, 40_000, 4), (9000, 1_200_000, 1),
                                        (700, 1_100_000, 2), (10, 0, 1)])
def test_item_parallel_row_reductions_match_torch(dev, n_rows, E, H):
    """csrc/segscan.hip: npi_seg_rowsum_ex (plain and through a map) and npi_gat_softmax_stats_ex (+ per-entry scores)
    against index_add_ / scatter_reduce in fp64 -- hub rows cut over thousands of items (64- and 256-entry items), rows
    that end exactly on an item boundary, empty rows at the end and in between, several heads; run-to-run bit identical."""
    from npi_gnn_amd import functional as NF
    n_cols = n_rows + 17
    side, _, _ = _sides(dev, n_rows, n_cols, E, seed=n_rows + E, hub_rows=(3, n_rows // 2) if E else (), empty_from=max(n_rows - 7, 1))
    import numpy as np
    nnz = int(side.rowptr[-1])
    rp = side.rowptr.cpu().numpy().astype(np.int64)
    has = torch.from_numpy(rp[1:] > rp[:-1])
    hn = has.numpy()

    def by_row(vals_np, op, empty):
        """per-row reduction of entry-ordered values on the HOST (the CSR is sorted by row: ufunc.reduceat over the starts of
        the non-empty rows, which are strictly increasing and end where the next non-empty row begins)"""
        out = np.full((n_rows,) + vals_np.shape[1:], empty, dtype=np.float64)
        if nnz:
            out[hn] = op.reduceat(vals_np.astype(np.float64), rp[:-1][hn], axis=0)
        return out
    g = torch.Generator().manual_seed(5)
    vals = torch.randn(max(side.nnz_max, 1), H, generator=g).to(dev)
    want = torch.from_numpy(by_row(vals[:nnz].cpu().numpy(), np.add, 0.0))
    got = NF.seg_rowsum(side, vals, H)
    scale = float(want.abs().max().clamp(min=1.0))
    tol = 2e-6 * scale * max(1.0, (E / max(n_rows, 1)) ** 0.5)
    assert float((got.cpu().double() - want).abs().max()) < tol
    assert torch.equal(got, NF.seg_rowsum(side, vals, H))
    perm = torch.randperm(max(side.nnz_max, 1), generator=g).to(dev).to(torch.int32)
    got_m = NF.seg_rowsum(side, vals, H, map_=perm)
    want_m = torch.from_numpy(by_row(vals[perm[:nnz].long()].cpu().numpy(), np.add, 0.0))
    assert float((got_m.cpu().double() - want_m).abs().max()) < tol
    # softmax statistics + scores
    a_row = (torch.randn(n_rows, H, generator=g) * 3).to(dev)
    a_col = (torch.randn(n_cols, H, generator=g) * 3).to(dev)
    m, s, e = NF.gat_softmax_stats(side, a_row, a_col, H, 0.2, want_scores=True)
    row = side.rowidx[:nnz].long()
    z = torch.nn.functional.leaky_relu(a_row[row] + a_col[side.col[:nnz].long()], 0.2)
    assert torch.equal(e[:nnz], z)
    zc = z.cpu().numpy()
    m_ref = torch.from_numpy(by_row(zc, np.maximum, 0.0)).float()
    s_ref = torch.from_numpy(by_row(np.exp(zc.astype(np.float64) - m_ref.numpy().astype(np.float64)[row.cpu().numpy()]), np.add, 0.0))
    assert torch.equal(m.cpu(), m_ref)
    assert torch.allclose(s.cpu().double(), s_ref, rtol=2e-5, atol=1e-6)
    m0, s0 = NF.gat_softmax_stats(side, a_row, a_col, H, 0.2)   # without the score output: the same statistics
    assert torch.equal(m0, m) and torch.equal(s0, s)


@pytest.mark.parametrize("n_rows,E,item,C", [(50, 300, 64, 64), (5000, 90_000, 64, 256), (9000, 1_200_000, 256, 256),
                                              (700, 1_100_000, 64, 128), (3000, 400_000, 256, 32), (10, 0, 64, 64)])
def test_fused_backward_pass_also_leaves_the_row_sums_of_dz(dev, n_rows, E, item, C):
    """npi_gat_backward_fused_heads(g_src_out=): the by-source row sums of dz from the lanes that compute dz (a segmented scan per 64
    entries inside the pass, rows cut by an item boundary through segscan.hip's chain kernel) against npi_seg_rowsum_ex over the
    dz the same launch wrote and against fp64 sums -- hub rows cut over thousands of items, rows ending exactly on a block or item
    boundary, empty rows, both item sizes; dz, d hfeat and the row scales bit-equal to the launch without the row sums."""
    from npi_gnn_amd import functional as NF
    from npi_gnn_amd import graph as NG
    import numpy as np
    n_cols = n_rows + 17
    g = torch.Generator().manual_seed(n_rows + E)
    key = torch.randint(0, max(n_rows - 7, 1), (max(E, 1),), generator=g)[:E]
    for k, r in enumerate((3, n_rows // 2) if E else ()):
        key[k * (E // 4): k * (E // 4) + E // 5] = r
    # rows that end exactly where a 64-entry block / an item ends: pad row 0 up to a multiple of the item size
    val = torch.randint(0, n_cols, (E,), generator=g)
    side = NG.build_side(key.to(dev), val.to(dev), n_rows, n_cols, False, 0, False, item=item)
    nnz = int(side.rowptr[-1])
    dout = torch.randn(n_cols, C, generator=g).to(dev)
    hrow = torch.randn(n_rows, C, generator=g).to(dev)
    tpack = torch.randn(n_cols, 4, generator=g)
    tpack[:, 2] = tpack[:, 2].abs() * 0.1 + 0.01                          # 1 / s > 0
    tpack[:, 1] = tpack[:, 1].abs() + 2.0                                 # the "row max": keeps exp(. - m) bounded
    tpack = tpack.to(dev)
    a_src = torch.randn(n_rows, generator=g).to(dev)
    sc = torch.empty(n_rows, device=dev) if C == 256 else None
    gs = torch.full((n_rows,), float("nan"), device=dev)
    dh, dz = NF.gat_backward_fused_packed(side, dout, None, hrow, C, tpack, a_src, 0.2, scales_out=sc, rowsum_out=gs)
    sc0 = torch.empty(n_rows, device=dev) if C == 256 else None
    dh0, dz0 = NF.gat_backward_fused_packed(side, dout, None, hrow, C, tpack, a_src, 0.2, scales_out=sc0)
    assert torch.equal(dh, dh0) and torch.equal(dz[:nnz], dz0[:nnz]) and (sc is None or torch.equal(sc, sc0))
    rp = side.rowptr.cpu().numpy().astype(np.int64)
    hn = rp[1:] > rp[:-1]
    want = np.zeros(n_rows)
    if nnz:
        want[hn] = np.add.reduceat(dz[:nnz].double().cpu().numpy(), rp[:-1][hn])
    scale = max(1.0, float(np.abs(want).max()))
    tol = 2e-6 * scale * max(1.0, (E / max(n_rows, 1)) ** 0.5)
    assert not bool(torch.isnan(gs).any())
    assert float(np.abs(gs.double().cpu().numpy() - want).max()) < tol
    if nnz:
        ref = NF.seg_rowsum(side, dz.view(-1, 1), 1).view(-1)
        assert float((gs - ref).abs().max()) < tol
    gs2 = torch.empty_like(gs)
    NF.gat_backward_fused_packed(side, dout, None, hrow, C, tpack, a_src, 0.2, rowsum_out=gs2)
    assert torch.equal(gs, gs2)                                           # fixed orders: run-to-run bit identical
    # the gathered table in two parts (the sharded layers' hub table + own rows): the same entries, the same sums
    cut = n_cols // 3
    gs3 = torch.empty_like(gs)
    dh3, dz3 = NF.gat_backward_fused_packed(side, dout[:cut].contiguous(), dout[cut:].contiguous(), hrow, C, tpack, a_src, 0.2,
                                            rowsum_out=gs3)
    assert torch.equal(dh3, dh) and torch.equal(dz3[:nnz], dz[:nnz]) and torch.equal(gs3, gs)


@pytest.mark.parametrize("N,H,C", [(1000, 1, 256), (4097, 4, 64), (300_000, 1, 256), (50, 1, 8), (2000, 2, 512), (777, 3, 100)])
def test_rowdot_and_bias_gradient_in_one_pass(dev, N, H, C):
    from npi_gnn_amd import functional as NF
    g = torch.Generator().manual_seed(N)
    a = torch.randn(N, H * C, generator=g).to(dev)
    b = torch.randn(N, H * C, generator=g).to(dev)
    bias = torch.randn(H * C, generator=g).to(dev)
    D, cs = NF.gat_rowdot_colsum(a, b, bias, H, C)
    want_D = (a.double().view(N, H, C) * (b.double() - bias.double()).view(N, H, C)).sum(-1)
    assert torch.allclose(D.double(), want_D, rtol=1e-5, atol=1e-4)
    assert torch.allclose(cs.double(), a.double().sum(0), rtol=1e-5, atol=1e-3 * max(1.0, N ** 0.5 / 30))
    D2, none = NF.gat_rowdot_colsum(a, b, None, H, C, want_colsum=False)
    assert none is None and torch.allclose(D2.double(), (a.double().view(N, H, C) * b.double().view(N, H, C)).sum(-1), rtol=1e-5, atol=1e-4)
    assert torch.equal(NF.gat_rowdot_colsum(a, b, bias, H, C)[1], cs)


@pytest.mark.parametrize("H,C", [(1, 256), (1, 64), (2, 32), (1, 300)])
def test_fused_relu_equals_relu_behind_the_layer(dev, H, C):
    """gat_conv(..., relu=True) == F.relu(gat_conv(...)): output and every gradient (the ReLU in the aggregation's row
    epilogue, its mask in the rowdot / bias-gradient pass; several heads and odd widths take a separate pass)."""
    N, E, Fi = 3000, 40000, 64
    ei, x, W, att, b, go = _case(N, E, Fi, H, C, seed=H * 100 + C)
    res = []
    for fused in (False, True):
        xd, Wd, ad, bd = (t.to(dev).requires_grad_(True) for t in (x, W, att, b))
        out = npi.gat_conv(xd, ei.to(dev), Wd, ad, bd, heads=H, relu=fused)
        if not fused:
            out = torch.relu(out)
        out.backward(go.to(dev))
        res.append((out.detach(), xd.grad, Wd.grad, ad.grad, bd.grad))
    assert bool((res[1][0] >= 0).all()) and float((res[1][0] == 0).float().mean()) > 0.2      # a real ReLU
    assert torch.equal(res[0][0], res[1][0])
    for a, c in zip(res[0][1:], res[1][1:]):
        assert torch.allclose(a, c, rtol=1e-5, atol=1e-6 * float(a.abs().max()))


@pytest.mark.parametrize("N,E,Fi,C", [(3000, 40000, 128, 256), (5000, 60000, 256, 64), (130, 900, 128, 32)])
def test_attention_terms_in_the_gemm_epilogue_equal_the_separate_pass(dev, N, E, Fi, C):
    """One head: the terms g_dst (x) att_dst + g_src (x) att_src of d hfeat are never added to it -- dX takes them in the
    store epilogue of its GEMM (rank 2), dW as an outer-product correction from x^T g, d att from the same pass over x
    (functional._GatConvFn._backward_rank2).  Same gradients as with the read-modify-write pass
    (Schedule.gat_rank2_epilogue = False) up to f32 rounding; the forward is untouched."""
    from npi_gnn_amd.schedule import DEFAULT
    ei, x, W, att, b, go = _case(N, E, Fi, 1, C, seed=N + C)
    res = []
    for flag in (False, True):                          # (min_rows = 0: the product takes this path from 100,000 rows on)
        sch = DEFAULT.but(gat_rank2_epilogue=flag, gat_rank2_min_rows=0)
        xd, Wd, ad, bd = (t.to(dev).requires_grad_(True) for t in (x, W, att, b))
        out = npi.gat_conv(xd, ei.to(dev), Wd, ad, bd, heads=1, schedule=sch)
        out.backward(go.to(dev))
        res.append((out.detach(), xd.grad, Wd.grad, ad.grad, bd.grad))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][4], res[1][4])
    for a, c in zip(res[0][1:4], res[1][1:4]):
        assert torch.allclose(a, c, rtol=1e-4, atol=2e-6 * float(a.abs().max()) * max(1.0, N ** 0.5 / 30))
    assert not torch.equal(res[0][1], res[1][1])          # (the two paths really are different arithmetic)


@pytest.mark.parametrize("K,C", [(128, 256), (256, 64), (178, 300), (4, 8)])
def test_rank2_helper_products_match_torch(dev, K, C):
    """npi_gat_rank2_cols (U = att W^T) and npi_gat_rank2_tail (dW += P^T att, datt = P W) against fp64 matmuls; the in-place
    update leaves a dW with a row pitch intact outside its columns; either output may be skipped."""
    from npi_gnn_amd import functional as NF
    g = torch.Generator().manual_seed(K + C)
    W, att, P = torch.randn(K, C, generator=g), torch.randn(2, C, generator=g), torch.randn(2, K, generator=g)
    dw0 = torch.randn(K, C, generator=g)
    U = NF.gat_rank2_cols(W.to(dev), att.to(dev)).cpu()
    assert torch.allclose(U.double(), att.double() @ W.double().t(), atol=1e-5 * C ** 0.5, rtol=1e-5)
    wide = torch.full((K, C + 5), 7.0, device=dev)
    wide[:, :C] = dw0.to(dev)
    dw = wide[:, :C]
    datt = NF.gat_rank2_tail(P.to(dev), W.to(dev), att.to(dev), dw, True).cpu()
    assert torch.allclose(datt.double(), P.double() @ W.double(), atol=1e-5 * K ** 0.5, rtol=1e-5)
    assert torch.allclose(dw.cpu().double(), dw0.double() + P.double().t() @ att.double(), atol=1e-5, rtol=1e-5)
    assert bool((wide[:, C:] == 7.0).all())
    assert NF.gat_rank2_tail(P.to(dev), W.to(dev), att.to(dev), None, False) is None
    only = NF.gat_rank2_tail(P.to(dev), W.to(dev), att.to(dev), None, True)
    assert torch.equal(only.cpu(), datt)


@pytest.mark.parametrize("freeze", [(), ("weight",), ("att",), ("weight", "att"), ("bias",)])
def test_rank2_path_with_frozen_parameters(dev, freeze):
    """GATConv at a size that takes the rank-2 store epilogue (>= 100,000 rows) with some parameters frozen: the gradients that
    are asked for equal the ones of the separate-pass backward, the others stay None (no stray work for a frozen weight / att)."""
    from npi_gnn_amd import functional as NF
    from npi_gnn_amd.synth import bipartite_edge_index
    N, E, F = 120_000, 1_200_000, 128
    ei = bipartite_edge_index(N, E, seed=5).to(dev)
    g = npi.CSRGraph(ei, N)
    gen = torch.Generator(device=dev).manual_seed(2)
    x0 = torch.randn(N, F, device=dev, generator=gen)
    go = torch.randn(N, F, device=dev, generator=gen)

    def run(rank2):
        from npi_gnn_amd.schedule import DEFAULT
        torch.manual_seed(0)
        conv = npi.GATConv(F, F, schedule=DEFAULT.but(gat_rank2_epilogue=rank2)).to(dev)
        for name in freeze:
            getattr(conv, name).requires_grad_(False)
        x = x0.clone().requires_grad_(True)
        conv(x, g).backward(go)
        return x.grad, conv.weight.grad, conv.att.grad, conv.bias.grad
    a, b = run(True), run(False)
    for name, p, q in zip(("dx", "dW", "datt", "db"), a, b):
        assert (p is None) == (q is None), name
        if p is not None:
            assert float((p - q).abs().max()) <= 1e-4 * float(q.abs().max()), name
    assert a[1] is None if "weight" in freeze else a[1] is not None


@pytest.mark.gpu
@pytest.mark.parametrize("M,K,N", [(128, 64, 128), (40_000, 64, 256), (1000, 256, 256), (130_001, 128, 256), (70_000, 160, 128)])
def test_scores_in_the_gemm_epilogue_equal_the_pass_over_h(dev, M, K, N):
    """npi_linear_fwd_scores: h = x W bit-equal to npi_linear_fwd, and both row dots within fp32 rounding of npi_gat_scores on
    that h (ragged last row tile, both tile widths, several tiles per workgroup); bitwise reproducible."""
    from npi_gnn_amd import functional as NF
    g = torch.Generator(device=dev).manual_seed(M + K)
    x = torch.randn(M, K, device=dev, generator=g)
    W = torch.randn(K, N, device=dev, generator=g) / K ** 0.5
    att = torch.randn(1, 2 * N, device=dev, generator=g)
    assert NF.linear_fwd_scores_ok(x, W)
    h, a_dst, a_src = NF.linear_fwd_scores(x, W, att)
    h_ref = NF.linear_fwd(x, W)
    d_ref, s_ref = NF.gat_scores(h_ref, att.view(1, 2 * N), 1, N)
    assert torch.equal(h, h_ref)
    ref64 = h_ref.double() @ att.view(2, N).double().t()
    scale = float(ref64.abs().max())
    assert float((a_dst.double().view(-1) - ref64[:, 0]).abs().max()) <= 2e-6 * scale
    assert float((a_src.double().view(-1) - ref64[:, 1]).abs().max()) <= 2e-6 * scale
    assert float((a_dst - d_ref).abs().max()) <= 4e-6 * scale and float((a_src - s_ref).abs().max()) <= 4e-6 * scale
    for _ in range(3):
        h2, d2, s2 = NF.linear_fwd_scores(x, W, att)
        assert torch.equal(h2, h) and torch.equal(d2, a_dst) and torch.equal(s2, a_src)
    assert not NF.linear_fwd_scores_ok(x, torch.randn(K, 192, device=dev))      # two column tiles: not served
    # K = 32 is two k-steps per tile: the epilogue's parity double buffer needs four -- refused, the layer then
    # takes the pass over h
    assert not NF.linear_fwd_scores_ok(x[:, :32].contiguous(), torch.randn(32, N, device=dev))


@pytest.mark.gpu
@pytest.mark.parametrize("N,H,C", [(100, 1, 256), (70_001, 1, 256), (5_000, 1, 128), (33_333, 1, 512), (9_000, 1, 1024), (4_097, 4, 64),
                                   (12_345, 2, 32), (3_000, 1, 64), (2_000, 1, 257), (2_000, 3, 20), (600_000, 1, 256)])
def test_att_grad_pass_matches_the_definition(dev, N, H, C):
    """npi_gat_att_grad: datt[h, :C] = sum_i g_dst[i, h] hfeat[i, h, :], datt[h, C:] likewise with g_src -- the 16-byte kernel
    (one head and wide rows: per-row scalars broadcast with v_readlane; several heads / narrow rows: per-lane scalars), its
    full 64-row blocks and ragged tails, the 4-byte fallback (odd widths), against fp64; bitwise reproducible."""
    from npi_gnn_amd import functional as NF
    g = torch.Generator(device=dev).manual_seed(N + 7 * H + C)
    h = torch.randn(N, H * C, device=dev, generator=g)
    gd = torch.randn(N, H, device=dev, generator=g)
    gs = torch.randn(N, H, device=dev, generator=g)
    got = NF.gat_att_grad(h, gd, gs, H, C)
    h64 = h.double().view(N, H, C)
    ref = torch.cat([(gd.double().unsqueeze(-1) * h64).sum(0), (gs.double().unsqueeze(-1) * h64).sum(0)], dim=1)      # [H, 2C]
    scale = float(ref.abs().max())
    assert got.shape == (H, 2 * C)
    assert float((got.double() - ref).abs().max()) <= 2e-5 * scale + 1e-6 * N ** 0.5
    assert torch.equal(NF.gat_att_grad(h, gd, gs, H, C), got)
    # a row pitch wider than the row (a view of a larger buffer)
    wide = torch.zeros(N, H * C + 12, device=dev)
    wide[:, : H * C] = h
    assert torch.equal(NF.gat_att_grad(wide[:, : H * C], gd, gs, H, C), got) or \
        float((NF.gat_att_grad(wide[:, : H * C], gd, gs, H, C) - got).abs().max()) <= 1e-5 * scale


@pytest.mark.gpu
@pytest.mark.parametrize("item", [64, 256])
@pytest.mark.parametrize("N,E,C,hub", [(3000, 40_000, 256, 0), (20_000, 600_000, 128, 300_000), (5_000, 150_000, 64, 70_000),
                                       (900, 9_000, 36, 0)])
def test_fused_forward_statistics_equal_the_statistics_pass(dev, N, E, C, hub, item):
    """npi_gat_aggregate_fused (round 5): the forward aggregation that computes the scores and the softmax statistics itself
    against the two-launch path (npi_gat_softmax_stats_ex + npi_gat_aggregate_scores): the row maxima EXACTLY, the row sums and the
    output to fp32 rounding (the parts of a cut row are merged with exp(m_part - m_row) in a fixed order) -- with a hub row cut
    over hundreds of workgroups (the two-level chain), both item sizes, scores that spread over +-20 (every merge rescales),
    bias + ReLU; the hub row against the formula in fp64; bitwise reproducible."""
    from npi_gnn_amd import functional as NF
    g = torch.Generator().manual_seed(N + C + item)
    ei = torch.randint(0, N, (2, E), generator=g)
    if hub:
        ei[1, :hub] = 7                                             # one target with `hub` in-edges: its row spans many workgroups
    graph = npi.CSRGraph(ei.to(dev), N, item=item)
    d = graph.by_dst
    h = torch.randn(N, C, generator=g).to(dev)
    att = (torch.randn(1, 2 * C, generator=g) * (6.0 / C ** 0.5)).to(dev)            # scores spread over about +-20
    a_dst, a_src = NF.gat_scores(h, att, 1, C)                        # <h_i, att[:C]>, <h_i, att[C:]> per node
    bias = torch.randn(C, generator=g).to(dev)
    m0, s0, sc = NF.gat_softmax_stats(d, a_dst, a_src, 1, 0.2, want_scores=True)
    ref = NF.gat_aggregate_scores(d, h, None, C, sc, m0, s0, bias=bias, relu=True)
    out, m, s = NF.gat_aggregate_fused(d, h, None, C, a_dst, att, 0.2, bias=bias, relu=True)
    torch.cuda.synchronize()
    # the fused launch recomputes the source half of every score from the gathered row: another summation order of the same
    # C-term dot, so maxima and sums agree to that rounding (relative to the scores' scale), not bit for bit
    tol = 4e-6 * max(1.0, float(sc.abs().max()))
    assert float((m - m0).abs().max()) <= tol
    assert float(((s - s0).abs() / s0.abs().clamp(min=1e-30)).max()) <= 10 * tol + 2e-5
    scale = float(ref.abs().max())
    assert float((out - ref).abs().max()) <= 5e-5 * max(scale, 1.0)
    for _ in range(2):
        o2, m2, s2 = NF.gat_aggregate_fused(d, h, None, C, a_dst, att, 0.2, bias=bias, relu=True)
        assert torch.equal(o2, out) and torch.equal(m2, m) and torch.equal(s2, s)
    # the heaviest row against the definition in fp64 (self loop included: add_self_loops after remove_self_loops)
    i = 7 if hub else int(torch.bincount(ei[1], minlength=N).argmax())
    src = ei[0][(ei[1] == i) & (ei[0] != i)].to(dev)
    src = torch.cat([src, torch.tensor([i], device=dev)])
    z = (a_dst[i, 0] + a_src[src, 0]).double()
    e = torch.where(z > 0, z, 0.2 * z)
    w = torch.softmax(e, 0)
    want = torch.relu((w.view(-1, 1) * h[src].double()).sum(0) + bias.double())
    assert float((out[i].double() - want).abs().max()) <= 1e-5 * max(1.0, float(want.abs().max()))
    assert abs(float(m[i, 0]) - float(e.max())) <= tol
    assert abs(float(s[i, 0]) - float((e - e.max()).exp().sum())) <= (10 * tol + 2e-5) * float((e - e.max()).exp().sum())


@pytest.mark.gpu
def test_fused_forward_on_an_edge_list_with_empty_rows_and_under_load(dev):
    """rows without any entry (a graph built without self loops) come out as bias with m = s = 0, as the statistics pass leaves
    them; 40 launches through one scratch buffer beside a second stream's load stay bit-identical (the arrival counters and the
    (m, s) slots of the partial rows are reused launch after launch)"""
    from npi_gnn_amd import functional as NF
    g = torch.Generator().manual_seed(5)
    N, E, C = 50_000, 1_500_000, 256
    ei = torch.randint(0, N // 2, (2, E), generator=g)              # the upper half of the nodes has no edge at all
    ei[1, :400_000] = 11
    graph = npi.CSRGraph(ei.to(dev), N, self_loops=False, keep_equal=True)
    d = graph.by_dst
    h = torch.randn(N, C, generator=g).to(dev)
    att = (torch.randn(1, 2 * C, generator=g) * (3.0 / C ** 0.5)).to(dev)
    a_dst, a_src = NF.gat_scores(h, att, 1, C)
    bias = torch.randn(C, generator=g).to(dev)
    out, m, s = NF.gat_aggregate_fused(d, h, None, C, a_dst, att, 0.2, bias=bias)
    m0, s0, sc = NF.gat_softmax_stats(d, a_dst, a_src, 1, 0.2, want_scores=True)
    ref = NF.gat_aggregate_scores(d, h, None, C, sc, m0, s0, bias=bias)
    assert float((m - m0).abs().max()) <= 4e-6 * max(1.0, float(sc.abs().max()))
    assert float((out - ref).abs().max()) <= 5e-5 * float(ref.abs().max())
    empty = torch.bincount(ei[1], minlength=N).to(dev) == 0
    assert bool(empty.any()) and torch.equal(out[empty], bias.expand(int(empty.sum()), C))
    assert float(m[empty].abs().max()) == 0.0 and float(s[empty].abs().max()) == 0.0
    side = torch.cuda.Stream(device=dev)
    big = torch.randn(64_000_000, device=dev)
    with torch.cuda.stream(side):
        for _ in range(20):
            big.mul_(1.0001)
    for k in range(40):
        o2, m2, s2 = NF.gat_aggregate_fused(d, h, None, C, a_dst, att, 0.2, bias=bias)
        assert torch.equal(o2, out) and torch.equal(s2, s), k
    torch.cuda.synchronize()
