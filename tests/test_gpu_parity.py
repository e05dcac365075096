"""Parity of the HIP path (through the C ABI) with the CPU oracle on the same seeded inputs.

Bar (north_star): node/edge indexing bit-exact; fp32 outputs within 1e-4 of the PyG-style CPU path.
Tolerances are written per test: values are O(1) so `atol=1e-4, rtol=1e-4` is the 1e-4 bar.
"""
import numpy as np
import pytest
import torch

import npi_gnn_amd as npi
from npi_gnn_amd import functional as NF
from npi_gnn_amd import graph as NG
from oracle import ref_conv as R

pytestmark = pytest.mark.gpu

ATOL = 1e-4
RTOL = 1e-4
from _util import GRAD_REL, rel_max  # noqa: E402


def rand_edges(N, E, seed, hub=None):
    g = torch.Generator().manual_seed(seed)
    ei = torch.randint(0, N, (2, E), generator=g)
    if hub is not None:                       # a few very heavy target rows (skew: reference max deg 1121)
        n_h = E // 3
        ei[1, :n_h] = hub
    return ei


def csr_reference(key, val, N, loops=True, sort_columns=False):
    """numpy restatement of the build contract: drop key==val, stable sort by key (``sort_columns``: by (key, column), ties in
    list order), loop last."""
    key, val = key.numpy(), val.numpy()
    keep = key != val
    eids = np.nonzero(keep)[0]
    order = np.lexsort((val[keep], key[keep])) if sort_columns else np.argsort(key[keep], kind="stable")
    k_s, v_s, e_s = key[keep][order], val[keep][order], eids[order]
    rows_col, rows_eid, rows_idx, rowptr = [], [], [], [0]
    bounds = np.searchsorted(k_s, np.arange(N + 1))
    for r in range(N):
        c = list(v_s[bounds[r]:bounds[r + 1]])
        e = list(e_s[bounds[r]:bounds[r + 1]])
        if loops:
            c.append(r)
            e.append(-1)
        rows_col += c
        rows_eid += e
        rows_idx += [r] * len(c)
        rowptr.append(len(rows_col))
    return (np.array(rowptr, np.int32), np.array(rows_col, np.int32), np.array(rows_eid, np.int32),
            np.array(rows_idx, np.int32))


@pytest.mark.parametrize("N,E,loops", [(1, 0, True), (5, 0, True), (7, 20, True), (300, 5000, True),
                                       (300, 5000, False), (5000, 200000, True), (70000, 300000, True),
                                       (3000, 1_100_000, True)])      # the last one: 256-entry items
def test_csr_build_is_bit_exact(dev, N, E, loops):
    ei = rand_edges(N, E, seed=N + E) if E else torch.zeros((2, 0), dtype=torch.long)
    g = npi.CSRGraph(ei.to(dev), N, self_loops=loops)
    for side, key, val in ((g.by_dst, ei[1], ei[0]), (g.by_src, ei[0], ei[1])):
        rowptr, col, eid, rowidx = csr_reference(key, val, N, loops)
        nnz = int(rowptr[-1])
        assert np.array_equal(side.rowptr.cpu().numpy(), rowptr)
        assert np.array_equal(side.col.cpu().numpy()[:nnz], col)
        assert np.array_equal(side.eid.cpu().numpy()[:nnz], eid)
        assert np.array_equal(side.rowidx.cpu().numpy()[:nnz], rowidx)
        # item_row[i] = row holding entry item_edges * i; the side carries its item size (the hint at build time: in this module a
        # stand-in that switches at 2^20 entries of capacity, tests/conftest.py)
        item_edges = side.item
        assert item_edges == NG.item_hint(side.nnz_max) == (64 if side.nnz_max < (1 << 20) else 256)
        assert side.n_items == -(-side.nnz_max // item_edges)
        ir = side.item_row.cpu().numpy()
        for i in range(1, side.n_items):
            k = item_edges * i
            if k < nnz:
                assert rowptr[ir[i]] <= k < rowptr[ir[i] + 1]
        assert int(side.status.item()) == 0


@pytest.mark.parametrize("N,E,loops", [(7, 20, True), (300, 5000, False), (5000, 200000, True), (3000, 1_100_000, True)])
def test_csr_build_with_sorted_columns_is_bit_exact(dev, N, E, loops):
    """NPI_CSR_SORT_COLUMNS: the same rows, every row's entries in column order (duplicate edges: list order), eid still the
    entry's position in the caller's list; an aggregation over it gives the sums of the plain build in another association"""
    ei = rand_edges(N, E, seed=N + E, hub=N // 2)               # a third of the edges share one target: long rows, duplicates
    g = npi.CSRGraph(ei.to(dev), N, self_loops=loops, sort_columns=True)
    for side, key, val in ((g.by_dst, ei[1], ei[0]), (g.by_src, ei[0], ei[1])):
        rowptr, col, eid, rowidx = csr_reference(key, val, N, loops, sort_columns=True)
        nnz = int(rowptr[-1])
        assert np.array_equal(side.rowptr.cpu().numpy(), rowptr)
        assert np.array_equal(side.col.cpu().numpy()[:nnz], col)
        assert np.array_equal(side.eid.cpu().numpy()[:nnz], eid)
        assert np.array_equal(side.rowidx.cpu().numpy()[:nnz], rowidx)
        assert int(side.status.item()) == 0
    plain = npi.CSRGraph(ei.to(dev), N, self_loops=loops)
    gen = torch.Generator().manual_seed(1)
    x = torch.randn(N, 64, generator=gen).to(dev)
    # (unweighted: with several self loops on one node in the list, WHICH of their weights becomes the node's loop weight is
    # as unspecified here as it is in PyG's add_remaining_self_loops on a GPU)
    a, b = (npi.sage_conv(x, gr, torch.eye(64, device=dev), None) for gr in (g, plain))
    assert float((a - b).abs().max()) <= 1e-5 * float(b.abs().max())
    # out-of-range ids are flagged and dropped exactly as by the plain build
    bad = torch.tensor([[0, 1, 9, 2], [1, 2, 0, -5]])
    gb = npi.CSRGraph(bad.to(dev), 3, sort_columns=True)
    assert int(gb.by_dst.status.item()) == 1 and gb.by_dst.rowptr.cpu().tolist() == [0, 1, 3, 5]
    with pytest.raises(IndexError):                              # ... and reported at the next device read
        gb.nnz()


def test_csr_build_flags_out_of_range_ids(dev):
    ei = torch.tensor([[0, 1, 9], [1, 2, 0]])
    g = npi.CSRGraph(ei.to(dev), 3)
    assert int(g.by_dst.status.item()) == 1
    assert g.by_dst.rowptr.cpu().tolist() == [0, 1, 3, 5]      # bad column dropped
    # a build never synchronises; the dropped id is reported at the next device read (PyG raises here)
    with pytest.raises(IndexError):
        g.nnz()
    ok = npi.CSRGraph(torch.tensor([[0, 1], [1, 2]]).to(dev), 3)
    assert ok.nnz() == 5                                       # the report is made once, later graphs are clean
    npi.set_debug(True)
    try:
        with pytest.raises(IndexError):
            npi.CSRGraph(ei.to(dev), 3)
    finally:
        npi.set_debug(False)


def test_segsum_on_a_csr_whose_every_edge_was_dropped(dev):
    """capacity nnz_max > 0 but rowptr[N] == 0 (self_loops=False and only self loops / bad ids in the edge
    list): every output row must still be written (zero, or the bias)."""
    N, F = 300, 256
    ei = torch.arange(N).repeat(2, 1)                          # nothing but self loops
    g = npi.CSRGraph(ei.to(dev), N, self_loops=False)
    x = torch.randn(N, F, device=dev)
    bias = torch.randn(F, device=dev)
    for mean in (False, True):
        out = torch.full((N, F), float("nan"), device=dev)
        NF.segsum(g, g.by_dst, x, mean=mean, out=out)
        assert bool((out == 0).all())
        out = torch.full((N, F), float("nan"), device=dev)
        NF.segsum(g, g.by_dst, x, mean=mean, bias=bias, out=out)
        assert torch.equal(out, bias.expand(N, F))
    xs = torch.randn(N, 64, device=dev)                        # the narrow-row kernel
    out = torch.full((N, 64), float("nan"), device=dev)
    NF.segsum(g, g.by_dst, xs, out=out)
    assert bool((out == 0).all())
    assert g.nnz() == 0


def _segsum_oracle(ei, N, x, w_entry_fn=None, mean=False, loops=True):
    ei2 = R.add_remaining_self_loops(ei, None, 1.0, N)[0] if loops else ei[:, ei[0] != ei[1]]
    msg = x.index_select(0, ei2[0])
    return (R.scatter_mean if mean else R.scatter_add)(msg, ei2[1], N)


@pytest.mark.parametrize("F", [1, 3, 64, 65, 128, 178, 256, 300, 512, 1100])
@pytest.mark.parametrize("mean", [False, True])
def test_segsum_widths(dev, F, mean):
    N, E = 700, 9000
    ei = rand_edges(N, E, seed=F, hub=5)       # row 5 has ~3000 entries: cut across ~12 items
    x = torch.randn(N, F, generator=torch.Generator().manual_seed(1))
    g = npi.CSRGraph(ei.to(dev), N)
    out = NF.segsum(g, g.by_dst, x.to(dev), mean=mean).cpu()
    ref = _segsum_oracle(ei, N, x.double(), mean=mean).float()
    assert torch.allclose(out, ref, atol=ATOL * (1 if mean else 30), rtol=RTOL)
    out_t = NF.segsum(g, g.by_src, x.to(dev), mean=mean).cpu()
    ref_t = _segsum_oracle(ei.flip(0), N, x.double(), mean=mean).float()
    assert torch.allclose(out_t, ref_t, atol=ATOL * (1 if mean else 30), rtol=RTOL)


@pytest.mark.parametrize("F", [256, 128, 64, 20])            # 128 / 64 / 20: the several-entries-per-instruction kernel
def test_segsum_empty_rows_and_no_self_loops(dev, F):
    N = 2000
    ei = rand_edges(600, 3000, seed=3) + 700           # rows [0,700) and [1300,2000) are empty
    x = torch.randn(N, F)
    g = npi.CSRGraph(ei.to(dev), N, self_loops=False)
    out = NF.segsum(g, g.by_dst, x.to(dev)).cpu()
    ref = _segsum_oracle(ei, N, x.double(), loops=False).float()
    assert torch.allclose(out, ref, atol=1e-3, rtol=RTOL)
    assert float(out[:700].abs().max()) == 0.0 and float(out[1300:].abs().max()) == 0.0


@pytest.mark.parametrize("F", [256, 128, 64, 20])
def test_segsum_weighted_and_bias(dev, F):
    N, E = 900, 20000
    ei = rand_edges(N, E, seed=11, hub=17)
    x = torch.randn(N, F)
    g = npi.CSRGraph(ei.to(dev), N)
    nnz = g.nnz()
    w = torch.rand(nnz)
    bias = torch.randn(F)
    out = NF.segsum(g, g.by_dst, x.to(dev), w=w.to(dev), bias=bias.to(dev)).cpu()
    col = g.by_dst.col.cpu().long()[:nnz]
    row = g.by_dst.rowidx.cpu().long()[:nnz]
    ref = torch.zeros(N, F, dtype=torch.float64).index_add_(0, row, w.double().view(-1, 1) * x.double()[col]) + bias.double()
    assert torch.allclose(out, ref.float(), atol=2e-3, rtol=RTOL)


@pytest.mark.parametrize("F", [256, 128, 64])
def test_segsum_is_bitwise_reproducible(dev, F):
    N, E = 3000, 100000
    ei = rand_edges(N, E, seed=5, hub=1)
    x = torch.randn(N, F).to(dev)
    g = npi.CSRGraph(ei.to(dev), N)
    a = NF.segsum(g, g.by_dst, x, mean=True)
    b = NF.segsum(g, g.by_dst, x, mean=True)
    g2 = npi.CSRGraph(ei.to(dev), N)
    c = NF.segsum(g2, g2.by_dst, x, mean=True)
    assert torch.equal(a, b) and torch.equal(a, c)


@pytest.mark.parametrize("M,K,N", [(1, 1, 1), (37, 178, 128), (300, 128, 128), (1000, 256, 256),
                                   (129, 65, 2), (513, 64, 300)])
def test_linear_fwd_bwd(dev, M, K, N):
    g = torch.Generator().manual_seed(M + K + N)
    a = torch.randn(M, K, generator=g)
    w = torch.randn(K, N, generator=g) / K ** 0.5
    b = torch.randn(N, generator=g)
    rs = torch.rand(M, generator=g) + 0.5
    dc = torch.randn(M, N, generator=g)
    out = NF.linear_fwd(a.to(dev), w.to(dev), b.to(dev)).cpu()
    assert torch.allclose(out, (a.double() @ w.double() + b.double()).float(), atol=ATOL, rtol=RTOL)
    out = NF.linear_fwd(a.to(dev), w.to(dev), b.to(dev), rowscale=rs.to(dev), relu=True).cpu()
    ref = torch.relu(rs.double().view(-1, 1) * (a.double() @ w.double()) + b.double()).float()
    assert torch.allclose(out, ref, atol=ATOL, rtol=RTOL)
    da = NF.linear_bwd_data(dc.to(dev), w.to(dev), rowscale=rs.to(dev)).cpu()
    ref = (rs.double().view(-1, 1) * (dc.double() @ w.double().t())).float()
    assert torch.allclose(da, ref, atol=ATOL * 4, rtol=RTOL)
    dw, db = NF.linear_bwd_weight(a.to(dev), dc.to(dev))
    assert torch.allclose(dw.cpu(), (a.double().t() @ dc.double()).float(), atol=ATOL * max(1, M ** 0.5), rtol=RTOL)
    assert torch.allclose(db.cpu(), dc.double().sum(0).float(), atol=ATOL * max(1, M ** 0.5), rtol=RTOL)
    assert torch.allclose(NF.colsum(dc.to(dev)).cpu(), dc.double().sum(0).float(), atol=ATOL * max(1, M ** 0.5), rtol=RTOL)


def test_linear_bwd_weight_large_m_split(dev):
    M, K, N = 200_000, 256, 256
    g = torch.Generator().manual_seed(0)
    a = torch.randn(M, K, generator=g)
    dc = torch.randn(M, N, generator=g)
    dw, db = NF.linear_bwd_weight(a.to(dev), dc.to(dev))
    ref = (a.double().t() @ dc.double())
    err = (dw.cpu().double() - ref).abs().max() / ref.abs().max()
    assert float(err) < 1e-5


@pytest.mark.parametrize("M,K,N", [(200_003, 256, 256), (5085, 128, 128), (70_000, 128, 256), (4096 + 7, 256, 128)])
@pytest.mark.parametrize("shared", [False, True])
def test_bf16_dw_matches_f32_reference(dev, M, K, N, shared):
    """bf16 STORAGE (BASELINE.json configs[1]): dW = a^T dC and db on the matrix-core dW kernel with a native bf16 producer (one
    plane, no split: the operands are bf16 values).  Reference: the fp64 product of the SAME bf16-rounded operands -- every
    product is exact in f32, so what is checked is the f32 accumulation over M rows plus the ONE rounding of the result to
    bf16 (2^-9 relative); run-to-run bitwise reproducible; both grid regimes."""
    g = torch.Generator().manual_seed(M + K + N)
    a = torch.randn(M, K, generator=g).bfloat16()
    dc = torch.randn(M, N, generator=g).bfloat16()
    dw, db = NF.linear_bwd_weight(a.to(dev), dc.to(dev), shared=shared)
    assert dw.dtype == torch.bfloat16 and db.dtype == torch.bfloat16
    ref_w = a.double().t() @ dc.double()
    ref_b = dc.double().sum(0)
    for got, ref in ((dw, ref_w), (db, ref_b)):
        got = got.double().cpu()
        # every element within one bf16 rounding (2^-9 relative; 2^-8 allowed) of the fp64 value, + f32 accumulation noise
        assert bool(((got - ref).abs() <= ref.abs() * 2.0 ** -8 + 1e-5 * float(ref.abs().max())).all())
    dw2, db2 = NF.linear_bwd_weight(a.to(dev), dc.to(dev), shared=shared)
    assert torch.equal(dw, dw2) and torch.equal(db, db2)
    # the f32 path on the widened operands agrees to bf16 rounding: the two kernels compute the same sums
    dwf, dbf = NF.linear_bwd_weight(a.to(dev).float(), dc.to(dev).float(), shared=shared)
    assert float((dwf.double().cpu() - ref_w).abs().max() / ref_w.abs().max()) <= 1e-5
    assert torch.equal(dwf.bfloat16(), dw) or float((dwf.bfloat16().float() - dw.float()).abs().max() / dwf.abs().max()) <= 2.0 ** -7


def _layer_case(N, E, Fi, Fo, seed, symmetric):
    ei = rand_edges(N, E, seed, hub=3)
    if symmetric:
        ei = torch.cat([ei, ei.flip(0)], dim=1)
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(N, Fi, generator=g)
    W = (torch.rand(Fi, Fo, generator=g) * 2 - 1) / Fi ** 0.5
    b = (torch.rand(Fo, generator=g) * 2 - 1) / Fi ** 0.5
    go = torch.randn(N, Fo, generator=g)
    return ei, x, W, b, go


@pytest.mark.parametrize("N,E,Fi,Fo,sym", [(50, 120, 178, 128, True), (4000, 30000, 128, 128, True),
                                           (4000, 30000, 256, 256, False), (2500, 9000, 65, 64, False),
                                           (30000, 1_100_000, 64, 64, False),     # >= 2^20 entries: 256-entry items
                                           (20000, 1_080_000, 128, 32, False)])
def test_sage_conv_fwd_bwd_matches_oracle(dev, N, E, Fi, Fo, sym):
    ei, x, W, b, go = _layer_case(N, E, Fi, Fo, 7, sym)
    ref_out, ref_dx, ref_dw, ref_db = R.sage_layer_fwd_bwd(x, ei, W, b, go)
    xd = x.to(dev).requires_grad_(True)
    Wd = W.to(dev).requires_grad_(True)
    bd = b.to(dev).requires_grad_(True)
    out = npi.sage_conv(xd, ei.to(dev), Wd, bd)
    out.backward(go.to(dev))
    assert torch.allclose(out.detach().cpu(), ref_out, atol=ATOL, rtol=RTOL)
    assert torch.allclose(xd.grad.cpu(), ref_dx, atol=ATOL, rtol=RTOL)
    # parameter gradients (sums over all N rows): max |diff| / max |ref| <= 1e-5 against the oracle evaluated in fp64
    _, _, dw64, db64 = R.sage_layer_fwd_bwd(x.double(), ei, W.double(), b.double(), go.double())
    assert rel_max(Wd.grad, dw64) <= GRAD_REL and rel_max(bd.grad, db64) <= GRAD_REL


def test_sage_conv_edge_weight_matches_oracle(dev):
    """`edge_weight.view(-1, 1) * x_j`, mean over the entry count; an existing self loop keeps its own weight as the
    loop weight (add_remaining_self_loops), every other node's loop weighs 1."""
    N, E, Fi, Fo = 700, 9000, 178, 128
    g = torch.Generator().manual_seed(21)
    ei = rand_edges(N, E, 21, hub=4)
    same = ei[0] == ei[1]
    ei[1, same] = (ei[1, same] + 1) % N                                        # no accidental (possibly duplicated) self loops
    ei[:, :5] = torch.tensor([[3, 9, 9, 20, 50], [3, 9, 10, 20, 51]])          # explicit self loops (node 9 also a plain edge)
    ew = torch.rand(E, generator=g) * 2.0
    x = torch.randn(N, Fi, generator=g)
    W, b, go = torch.randn(Fi, Fo, generator=g) / Fi ** 0.5, torch.randn(Fo, generator=g), torch.randn(N, Fo, generator=g)
    xr, Wr, br = (t.clone().requires_grad_(True) for t in (x, W, b))
    ref = R.sage_conv(xr, ei, Wr, br, edge_weight=ew)
    ref.backward(go)
    conv = npi.SAGEConv(Fi, Fo).to(dev)
    with torch.no_grad():
        conv.weight.copy_(W)
        conv.bias.copy_(b)
    xd = x.to(dev).requires_grad_(True)
    out = conv(xd, ei.to(dev), edge_weight=ew.to(dev))
    out.backward(go.to(dev))
    assert torch.allclose(out.detach().cpu(), ref.detach(), atol=ATOL, rtol=RTOL)
    assert torch.allclose(xd.grad.cpu(), xr.grad, atol=ATOL, rtol=RTOL)
    x6, W6, b6 = (t.double().clone().requires_grad_(True) for t in (x, W, b))
    R.sage_conv(x6, ei, W6, b6, edge_weight=ew.double()).backward(go.double())
    assert rel_max(conv.weight.grad, W6.grad) <= GRAD_REL and rel_max(conv.bias.grad, b6.grad) <= GRAD_REL
    # without weights the same module call is the reference's plain layer
    plain = conv(x.to(dev), ei.to(dev)).detach().cpu()
    assert torch.allclose(plain, R.sage_conv(x, ei, W, b).detach(), atol=ATOL, rtol=RTOL)


def test_sage_direction_on_directed_toy(dev):
    x = torch.tensor([[1.0, 10.0], [3.0, 30.0], [5.0, 50.0]])
    ei = torch.tensor([[0, 0], [1, 2]])                      # 0->1, 0->2
    W = torch.eye(2)
    out = npi.sage_conv(x.to(dev), ei.to(dev), W.to(dev), None).cpu()
    assert out.tolist() == [[1.0, 10.0], [2.0, 20.0], [3.0, 30.0]]


@pytest.mark.parametrize("Fi,Fo", [(178, 256), (178, 64)])      # (A x) W when F_in <= F_out, A (x W) otherwise
@pytest.mark.parametrize("weighted,improved", [(False, False), (True, False), (False, True)])
def test_gcn_conv_fwd_bwd_matches_oracle(dev, weighted, improved, Fi, Fo):
    N, E = 3000, 25000
    ei, x, W, b, go = _layer_case(N, E, Fi, Fo, 9, symmetric=not weighted)
    ew = torch.rand(ei.size(1), generator=torch.Generator().manual_seed(1)) + 0.1 if weighted else None
    xr = x.clone().requires_grad_(True)
    Wr = W.clone().requires_grad_(True)
    br = b.clone().requires_grad_(True)
    ref = R.gcn_conv(xr, ei, Wr, br, ew, improved)
    ref.backward(go)
    xd = x.to(dev).requires_grad_(True)
    Wd = W.to(dev).requires_grad_(True)
    bd = b.to(dev).requires_grad_(True)
    out = npi.gcn_conv(xd, ei.to(dev), Wd, bd, None if ew is None else ew.to(dev), improved)
    out.backward(go.to(dev))
    assert torch.allclose(out.detach().cpu(), ref.detach(), atol=ATOL, rtol=RTOL)
    assert torch.allclose(xd.grad.cpu(), xr.grad, atol=ATOL, rtol=RTOL)
    x6, W6, b6 = (t.double().clone().requires_grad_(True) for t in (x, W, b))
    R.gcn_conv(x6, ei, W6, b6, None if ew is None else ew.double(), improved).backward(go.double())
    assert rel_max(Wd.grad, W6.grad) <= GRAD_REL and rel_max(bd.grad, b6.grad) <= GRAD_REL


@pytest.mark.parametrize("Fi,Fo,weighted", [(64, 128, True), (178, 64, True), (128, 128, False)])
def test_gcn_conv_normalize_false_matches_oracle(dev, Fi, Fo, weighted):
    """``GCNConv(normalize=False)`` (PyG 1.4.2): ``norm = edge_weight`` over the edge list as it is -- no self loop is added,
    existing (i, i) columns are ordinary messages; both evaluation orders.  Parity unpinned (the reference never constructs a
    GCNConv)."""
    N, E = 2500, 20000
    ei = rand_edges(N, E, seed=13, hub=3)
    ei[:, :2] = torch.tensor([[4, 9], [4, 9]])
    g = torch.Generator().manual_seed(14)
    x, go = torch.randn(N, Fi, generator=g), torch.randn(N, Fo, generator=g)
    ew = torch.rand(E, generator=g) + 0.1 if weighted else None
    conv = npi.GCNConv(Fi, Fo, normalize=False).to(dev)
    with torch.no_grad():
        conv.bias.uniform_(-0.5, 0.5)
    xd = x.to(dev).requires_grad_(True)
    out = conv(xd, ei.to(dev), None if ew is None else ew.to(dev))
    out.backward(go.to(dev))
    x6, W6, b6 = (t.detach().cpu().double().clone().requires_grad_(True) for t in (x, conv.weight, conv.bias))
    ref = R.gcn_conv(x6, ei, W6, b6, None if ew is None else ew.double(), normalize=False)
    ref.backward(go.double())
    scale = max(1.0, float(ref.abs().max()))
    assert float((out.detach().cpu().double() - ref.detach()).abs().max()) <= ATOL * scale
    assert float((xd.grad.cpu().double() - x6.grad).abs().max()) <= ATOL * max(1.0, float(x6.grad.abs().max()))
    assert rel_max(conv.weight.grad, W6.grad) <= GRAD_REL and rel_max(conv.bias.grad, b6.grad) <= GRAD_REL


def test_gat_conv_dropout_is_the_identity_in_evaluation_and_a_fresh_mask_per_training_call(dev):
    """attention dropout is the identity in evaluation (the reference's test loop calls model.eval()): the layer then equals the
    dropout-free one.  In training every call draws a fresh mask (the composed variant, functional._GatDropoutFn; its numbers
    against the oracle under the same mask: tests/test_gpu_gat.py), is differentiable, and keeps the expectation of the output."""
    ei = rand_edges(300, 2000, seed=3).to(dev)
    x = torch.randn(300, 32, device=dev)
    a, b = npi.GATConv(32, 16, heads=2, dropout=0.6).to(dev), npi.GATConv(32, 16, heads=2).to(dev)
    b.load_state_dict(a.state_dict())
    a.eval()
    assert torch.equal(a(x, ei), b(x, ei))
    a.train()
    torch.manual_seed(5)
    o1, o2 = a(x, ei), a(x, ei)
    assert o1.shape == o2.shape == (300, 32) and not torch.equal(o1, o2)
    o1.sum().backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in a.parameters())
    mean = torch.stack([a(x, ei).detach() for _ in range(400)]).mean(0)           # E[alpha keep] = alpha
    ref = b(x, ei).detach()
    assert float((mean - ref).abs().mean()) <= 0.12 * float(ref.abs().mean())    # (observed 0.06 at 300 draws, p = 0.6)


def test_modules_drop_into_a_net1_style_stack(dev):
    """conv -> relu -> conv, loss.backward(), optimizer.step(): the reference's call pattern
    (src/classes.py:62-70, src/train_with_twoDataset.PY:46-57) with the module interface."""
    torch.manual_seed(0)
    N, E = 1500, 6000
    ei = rand_edges(N, E, 1)
    ei = torch.cat([ei, ei.flip(0)], 1).to(dev)
    x = torch.randn(N, 178, device=dev)
    c1, c2 = npi.SAGEConv(178, 128).to(dev), npi.SAGEConv(128, 128).to(dev)
    opt = torch.optim.Adam(list(c1.parameters()) + list(c2.parameters()), lr=1e-3, weight_decay=1e-3)
    graph = npi.CSRGraph(ei, N)
    h = torch.relu(c2(torch.relu(c1(x, ei)), graph))          # edge_index or prebuilt graph
    sd1 = {k: v.detach().cpu().clone() for k, v in c1.state_dict().items()}
    sd2 = {k: v.detach().cpu().clone() for k, v in c2.state_dict().items()}
    ref = torch.relu(R.sage_conv(torch.relu(R.sage_conv(x.cpu(), ei.cpu(), sd1["weight"], sd1["bias"])),
                                 ei.cpu(), sd2["weight"], sd2["bias"]))
    assert torch.allclose(h.detach().cpu(), ref, atol=ATOL, rtol=RTOL)
    loss = h.pow(2).mean()
    loss.backward()
    opt.step()
    assert c1.weight.grad is not None and torch.isfinite(c1.weight.grad).all()
    assert not torch.equal(c1.weight.detach().cpu(), sd1["weight"])


@pytest.mark.parametrize("hubs", [False, True])
@pytest.mark.parametrize("world", [1, 3, 8])
def test_sharded_backend_virtual_ranks_on_one_gpu(dev, world, hubs):
    """with one GPU, run the W shards one after the other and emulate the all-gather of
    hub rows and the reduce-scatter of partial hub sums, to validate the sharded CSR sides (A: local rows
    x [hub table ; local light rows], B: hub rows x local rows) in both aggregation directions."""
    from npi_gnn_amd import dist as ND
    from npi_gnn_amd.synth import bipartite_edge_index, protein_mask
    F = 256
    if hubs:
        N = 5003
        ei = bipartite_edge_index(N, 60000, seed=21)
        mask = protein_mask(N)
    else:
        N = 5003
        ei = rand_edges(N, 60000, seed=21, hub=7)
        mask = None
    x = torch.randn(N, F, generator=torch.Generator().manual_seed(2))
    ref_fwd = R.sage_aggregate(x, ei)
    ei2 = R.add_remaining_self_loops(ei, None, 1.0, N)[0]
    ref_bwd = R.scatter_add(x.index_select(0, ei2[1]), ei2[0], N)      # transpose aggregation (sum)
    sgs = [ND.ShardedGraph(ei, N, r, world, dev, hub_mask=mask) for r in range(world)]
    part = sgs[0].part
    own = [sg.shard(x).to(dev) for sg in sgs]
    hub_table = torch.zeros(part.hub_rows, F, device=dev)
    for r, sg in enumerate(sgs):
        hub_table[r * part.h_per: r * part.h_per + sg.nH] = own[r][sg.nL:]
    be = sgs[0].backend
    for side_a, side_b, mean, ref in (("A", "B", True, ref_fwd), ("At", "Bt", False, ref_bwd)):
        partial = [be.segsum(getattr(sg, side_b), own[r]) for r, sg in enumerate(sgs)] if sgs[0].exchange_partials else None
        outs = []
        for r, sg in enumerate(sgs):
            # two-part table: [gathered hub rows ; the rank's own rows] (npi_segsum_ex), nothing copied
            agg = be.segsum(getattr(sg, side_a), hub_table, mean=mean, table2=own[r])
            if partial is not None and sg.nH:
                hsum = sum(p[r * part.h_per: r * part.h_per + sg.nH] for p in partial)
                if mean:
                    agg[sg.nL:] = (agg[sg.nL:] * sg.cnt_a_hub + hsum) * sg.inv_cnt[sg.nL:].view(-1, 1)
                else:
                    agg[sg.nL:] += hsum
            outs.append(agg.cpu())
        got = part.unshard(outs)
        assert torch.allclose(got, ref, atol=ATOL * (1 if mean else 30), rtol=RTOL)
    cnt = torch.bincount(ei2[1], minlength=N).float()
    for sg in sgs:
        assert torch.allclose(sg.inv_cnt.cpu(), 1.0 / cnt[sg.own.cpu()])


def test_sharded_layer_world1_matches_single_gpu_layer(dev):
    from npi_gnn_amd import dist as ND
    ei, x, W, b, go = _layer_case(3000, 20000, 256, 256, 5, True)
    sg = ND.ShardedGraph(ei, 3000, 0, 1, dev)
    layer = ND.ShardedSAGELayer(sg, W.to(dev), b.to(dev))
    xl = x.to(dev).requires_grad_(True)
    out = layer(xl)
    out.backward(go.to(dev))
    ref_out, ref_dx, ref_dw, ref_db = R.sage_layer_fwd_bwd(x, ei, W, b, go)
    assert torch.allclose(out.detach().cpu(), ref_out, atol=ATOL, rtol=RTOL)
    assert torch.allclose(xl.grad.cpu(), ref_dx, atol=ATOL, rtol=RTOL)
    _, _, dw64, db64 = R.sage_layer_fwd_bwd(x.double(), ei, W.double(), b.double(), go.double())
    assert rel_max(layer.weight.grad, dw64) <= GRAD_REL and rel_max(layer.bias.grad, db64) <= GRAD_REL


@pytest.mark.parametrize("M,K,N", [(1000, 256, 256), (128 * 5 + 3, 128, 384), (4096, 64, 128), (130, 256, 128)])
def test_split_bf16_gemm_is_f32_accurate(dev, M, K, N):
    """The default arithmetic (3-way bf16 split on the bf16 matrix cores) against an fp64 product: its error must
    be at the level of the exact-f32 kernel's (NPI_GEMM_EXACT_F32), through fwd (bias, rowscale, relu) and bwd_data,
    including the seam between the split interior tiles and the exact ragged strip."""
    g = torch.Generator().manual_seed(M + K)
    A = torch.randn(M, K, generator=g).to(dev)
    W = (torch.randn(K, N, generator=g) / K ** 0.5).to(dev)
    b = torch.randn(N, generator=g).to(dev)
    rs = torch.rand(M, generator=g).to(dev)
    dC = torch.randn(M, N, generator=g).to(dev)
    ref_f = torch.relu(rs.double().view(-1, 1) * (A.double() @ W.double()) + b.double())
    ref_b = rs.double().view(-1, 1) * (dC.double() @ W.double().t())
    from npi_gnn_amd._lib import NPI_GEMM_EXACT_F32, NPI_GEMM_SPLIT_BF16
    err = {}
    for mode, flags in ((0, NPI_GEMM_EXACT_F32), (1, NPI_GEMM_SPLIT_BF16)):          # per-call: no process-wide switch
        cf = NF.linear_fwd(A, W, b, rowscale=rs, relu=True, flags=flags)
        cb = NF.linear_bwd_data(dC, W, rs, flags=flags)
        err[mode] = (float((cf.double() - ref_f).abs().max() / ref_f.abs().max()),
                     float((cb.double() - ref_b).abs().max() / ref_b.abs().max()))
        again = NF.linear_fwd(A, W, b, rowscale=rs, relu=True, flags=flags)
        assert torch.equal(cf, again)                      # run-to-run bitwise reproducible
    for e0, e1 in zip(err[0], err[1]):
        assert e1 < 2e-6 and e1 < 3 * e0 + 1e-7, err


def test_non_finite_operands_of_the_projection_gemms(dev):
    """Pins the documented Inf / NaN behaviour of the two f32 arithmetics (include/npi_gnn.h at NPI_GEMM_*; reference:
    torch.matmul in PyG 1.4.2 SAGEConv.update).  Exact f32: an Inf operand gives Inf where torch.matmul does.  3-way
    bf16 split (the default): the same element gives NaN -- in the whole output row for an element of A, the whole output
    column for an element of W -- and every row / column that holds no such operand is untouched."""
    from npi_gnn_amd._lib import NPI_GEMM_EXACT_F32, NPI_GEMM_SPLIT_BF16
    g = torch.Generator().manual_seed(77)
    M, K, N = 1024, 256, 256
    A = torch.randn(M, K, generator=g).abs()                 # positive operands: an Inf product cannot meet an opposing Inf
    W = (torch.randn(K, N, generator=g).abs() / K ** 0.5)
    A[5, 17] = float("inf")                                   # one element of A
    A[9, 3] = 3.4e38                                          # finite in f32, beyond the largest bf16 (3.3895e38)
    A[300, 40] = float("nan")
    Ad, Wd = A.to(dev), W.to(dev)
    ref = A @ W                                               # fp32 matmul on the host = the reference's behaviour
    assert torch.isinf(ref[5]).all() and torch.isnan(ref[300]).all() and torch.isfinite(ref[9]).all()   # 3.4e38 * w, |w| << 1
    clean = torch.ones(M, dtype=torch.bool)
    clean[[5, 9, 300]] = False
    exact = NF.linear_fwd(Ad, Wd, None, flags=NPI_GEMM_EXACT_F32).cpu()
    split = NF.linear_fwd(Ad, Wd, None, flags=NPI_GEMM_SPLIT_BF16).cpu()
    for out in (exact, split):                                # rows without a non-finite operand: the ordinary accuracy
        assert torch.isfinite(out[clean]).all()
        assert float((out[clean] - ref[clean]).abs().max() / ref[clean].abs().max()) < 2e-6
    assert torch.isinf(exact[5]).all() and (exact[5] > 0).all()              # exact f32: Inf, like torch.matmul
    assert torch.isnan(exact[300]).all()
    assert float((exact[9] - ref[9]).abs().max() / ref[9].abs().max()) < 2e-6   # exact f32: finite, like torch.matmul
    assert torch.isnan(split[5]).all()                                        # split: Inf - bf16(Inf) = NaN poisons the row
    assert torch.isnan(split[9]).all()                                        # |x| beyond bf16's range behaves like Inf
    assert torch.isnan(split[300]).all()
    # an Inf in W: one output column
    W2 = W.clone()
    W2[11, 200] = float("inf")
    A2 = torch.randn(M, K, generator=g).abs().to(dev)
    e2 = NF.linear_fwd(A2, W2.to(dev), None, flags=NPI_GEMM_EXACT_F32).cpu()
    s2 = NF.linear_fwd(A2, W2.to(dev), None, flags=NPI_GEMM_SPLIT_BF16).cpu()
    cols = torch.ones(N, dtype=torch.bool)
    cols[200] = False
    assert torch.isinf(e2[:, 200]).all() and torch.isnan(s2[:, 200]).all()
    assert torch.isfinite(e2[:, cols]).all() and torch.isfinite(s2[:, cols]).all()
    # backward-data and dW follow the same rule (dC with an Inf element)
    dC = torch.randn(M, N, generator=g).abs()
    dC[7, 1] = float("inf")
    for flags, bad in ((NPI_GEMM_EXACT_F32, torch.isinf), (NPI_GEMM_SPLIT_BF16, torch.isnan)):
        da = NF.linear_bwd_data(dC.to(dev), Wd, None, flags=flags).cpu()
        assert bad(da[7]).all() and torch.isfinite(da[torch.arange(M) != 7]).all()
        # dW: the split kernel serves large node counts only (this M takes the exact one under either flag), so the
        # column is non-finite -- Inf or NaN -- and every other column is untouched
        dw, db = NF.linear_bwd_weight(A2, dC.to(dev), True, flags=flags)
        assert (~torch.isfinite(dw.cpu()[:, 1])).all() and torch.isfinite(dw.cpu()[:, 2:]).all() and torch.isfinite(dw.cpu()[:, 0]).all()
        assert not torch.isfinite(db.cpu()[1])
    Mb = 65536                                                    # large enough for gemm_dw_split_kernel
    Ab, dCb = torch.randn(Mb, K, generator=g).abs().to(dev), torch.randn(Mb, N, generator=g).abs()
    dCb[4097, 1] = float("inf")
    dwe = NF.linear_bwd_weight(Ab, dCb.to(dev), False, flags=NPI_GEMM_EXACT_F32)[0].cpu()
    dws = NF.linear_bwd_weight(Ab, dCb.to(dev), False, flags=NPI_GEMM_SPLIT_BF16)[0].cpu()
    assert torch.isinf(dwe[:, 1]).all() and torch.isnan(dws[:, 1]).all()
    assert torch.isfinite(dwe[:, 2:]).all() and torch.isfinite(dws[:, 2:]).all()


@pytest.mark.parametrize("M,K,N", [(4096, 256, 256), (100003, 128, 128), (50000 + 7, 256, 128), (20000, 128, 256),
                                   (300000, 256, 256), (8192 + 15, 384, 512)])
@pytest.mark.parametrize("shared", [False, True])
def test_split_bf16_dw_is_f32_accurate(dev, M, K, N, shared):
    """dW = A^T dC with BOTH operands split 3-way into bf16 on the fly (gemm_dw_split_kernel) against an fp64 product: the
    error must be at the level of the exact-f32 MFMA kernel's, for dW and for the fused column sums db, incl. node counts
    that are not a multiple of 16 (guarded remainder slab) and both grid regimes; run-to-run bit identical."""
    from npi_gnn_amd._lib import NPI_GEMM_EXACT_F32, NPI_GEMM_SPLIT_BF16
    g = torch.Generator().manual_seed(M + K + N)
    A = torch.randn(M, K, generator=g).to(dev)
    dC = (torch.randn(M, N, generator=g) * (1 + torch.rand(1, N, generator=g) * 3)).to(dev)
    ref_w = A.double().t() @ dC.double()
    ref_b = dC.double().sum(0)
    err = {}
    for mode, flags in ((0, NPI_GEMM_EXACT_F32), (1, NPI_GEMM_SPLIT_BF16)):
        dw, db = NF.linear_bwd_weight(A, dC, True, shared=shared, flags=flags)
        err[mode] = (float((dw.double() - ref_w).abs().max() / ref_w.abs().max()),
                     float((db.double() - ref_b).abs().max() / ref_b.abs().max()))
        dw2, db2 = NF.linear_bwd_weight(A, dC, True, shared=shared, flags=flags)
        assert torch.equal(dw, dw2) and torch.equal(db, db2)
        dw3, none = NF.linear_bwd_weight(A, dC, False, shared=shared, flags=flags)
        assert none is None and torch.equal(dw, dw3)
    for e0, e1 in zip(err[0], err[1]):
        assert e1 < 5e-6 and e1 < 3 * e0 + 2e-7, err


def test_gemm_entry_points_are_independent_across_threads_and_streams(dev):
    """the arithmetic and the dW grid regime are per-call arguments now.  Two threads, each on its own
    HIP stream, hammer the three GEMMs with DIFFERENT settings at the same time; every result must be bit-identical to
    the one the same call gives alone."""
    import threading
    from npi_gnn_amd._lib import NPI_GEMM_EXACT_F32, NPI_GEMM_SPLIT_BF16
    g = torch.Generator().manual_seed(5)
    M, K, N = 4096, 256, 256
    A = torch.randn(M, K, generator=g).to(dev)
    W = (torch.randn(K, N, generator=g) / K ** 0.5).to(dev)
    dC = torch.randn(M, N, generator=g).to(dev)
    settings = [(NPI_GEMM_EXACT_F32, False), (NPI_GEMM_SPLIT_BF16, True)]

    def run(flags, shared):
        return (NF.linear_fwd(A, W, None, flags=flags), NF.linear_bwd_data(dC, W, None, flags=flags)) + \
            NF.linear_bwd_weight(A, dC, True, shared=shared)
    alone = [tuple(t.clone() for t in run(*s)) for s in settings]
    torch.cuda.synchronize()
    assert not torch.equal(alone[0][0], alone[1][0])          # the two arithmetics do differ in the last bits
    bad = []

    def worker(k):
        try:
            st = torch.cuda.Stream(device=dev)
            with torch.cuda.stream(st):
                for _ in range(40):
                    got = run(*settings[k])
                    st.synchronize()
                    if not all(torch.equal(a, b) for a, b in zip(got, alone[k])):
                        bad.append(k)
        except Exception as e:                              # noqa: BLE001
            bad.append(repr(e))
    th = [threading.Thread(target=worker, args=(k,)) for k in (0, 1)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not bad, bad


def test_gcn_bf16_storage_trains(dev):
    """GCNConv(...).to(bfloat16) backward (bf16 storage, f32 accumulation) against the fp32 oracle on
    the bf16-rounded inputs."""
    N, E, Fi, Fo = 3000, 24000, 128, 128
    ei = rand_edges(N, E, seed=4, hub=5)
    g = torch.Generator().manual_seed(6)
    x, W, b, go = (torch.randn(N, Fi, generator=g), torch.randn(Fi, Fo, generator=g) / Fi ** 0.5,
                   torch.randn(Fo, generator=g), torch.randn(N, Fo, generator=g))
    xr, Wr, br, gor = (t.to(torch.bfloat16).float().requires_grad_(q) for t, q in ((x, True), (W, True), (b, True), (go, False)))
    ref = R.gcn_conv(xr, ei, Wr, br)
    ref.backward(gor)
    conv = npi.GCNConv(Fi, Fo).to(dev)
    with torch.no_grad():
        conv.weight.copy_(W)
        conv.bias.copy_(b)
    conv = conv.to(torch.bfloat16)
    xd = x.to(dev).to(torch.bfloat16).requires_grad_(True)
    out = conv(xd, ei.to(dev))
    out.backward(go.to(dev).to(torch.bfloat16))
    assert out.dtype == torch.bfloat16 and conv.weight.grad.dtype == torch.bfloat16

    def close(a, r, rel=2e-2):
        return float((a.float().cpu() - r).abs().max()) <= rel * float(r.abs().max())
    assert close(out.detach(), ref.detach()) and close(xd.grad, xr.grad) and close(conv.weight.grad, Wr.grad)
    assert close(conv.bias.grad, br.grad)


def test_training_step_is_hip_graph_capturable(dev):
    """No kernel on the path synchronises, allocates outside the stream order or reads sizes back: a full-batch
    fwd + bwd on a fixed graph captures into one HIP graph and replays bit-identically."""
    N, E, F = 3000, 20000, 128
    ei = rand_edges(N, E, seed=9, hub=3).to(dev)
    graph = npi.CSRGraph(ei, N)
    _ = graph.by_src
    torch.manual_seed(0)
    convs = torch.nn.ModuleList([npi.SAGEConv(F, F), npi.SAGEConv(F, F)]).to(dev)
    x = torch.randn(N, F, device=dev)

    def step():
        for p in convs.parameters():
            p.grad = None
        h = x
        for c in convs:
            h = torch.relu(c(h, graph))
        loss = h.pow(2).mean()
        loss.backward()
        return loss

    ref_loss = step().detach().clone()
    ref = [p.grad.clone() for p in convs.parameters()]
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        step()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    for p in convs.parameters():
        p.grad = None
    with torch.cuda.graph(g):
        static_loss = step()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    assert torch.equal(static_loss.detach(), ref_loss)
    for p, r in zip(convs.parameters(), ref):
        assert torch.equal(p.grad, r)


@pytest.mark.parametrize("M,K,N", [(1000, 256, 256), (128 * 5 + 3, 128, 384), (4096, 64, 128), (130, 192, 128), (777, 256, 512)])
def test_bf16_mfma_gemm_matches_f32_reference(dev, M, K, N):
    """bf16 storage: fwd (bias, rowscale, relu) and bwd_data through gemm_bf16_ws_kernel (K % 64 == 0) against an f32
    product of the SAME bf16-rounded inputs -- the only difference allowed is the rounding of the bf16 output."""
    g = torch.Generator().manual_seed(M + K + N)
    A = torch.randn(M, K, generator=g).to(torch.bfloat16)
    W = (torch.randn(K, N, generator=g) / K ** 0.5).to(torch.bfloat16)
    b = torch.randn(N, generator=g).to(torch.bfloat16)
    rs = torch.rand(M, generator=g)
    dC = torch.randn(M, N, generator=g).to(torch.bfloat16)
    ref_f = torch.relu(rs.view(-1, 1) * (A.float() @ W.float()) + b.float())
    ref_b = rs.view(-1, 1) * (dC.float() @ W.float().t())
    cf = NF.linear_fwd(A.to(dev), W.to(dev), b.to(dev), rowscale=rs.to(dev), relu=True)
    cb = NF.linear_bwd_data(dC.to(dev), W.to(dev), rs.to(dev))
    assert cf.dtype == torch.bfloat16 and cb.dtype == torch.bfloat16
    for got, ref in ((cf, ref_f), (cb, ref_b)):
        err = (got.float().cpu() - ref).abs()
        assert float((err / (ref.abs() + 1.0)).max()) < 1e-2           # one bf16 rounding of the result (2^-8)
        assert float(err.mean()) < 2e-3 * max(1.0, float(ref.abs().mean()))
    again = NF.linear_fwd(A.to(dev), W.to(dev), b.to(dev), rowscale=rs.to(dev), relu=True)
    assert torch.equal(cf, again)


def test_symmetric_edge_list_skips_the_second_sort(dev):
    """An edge list whose producer vouches for both directions (GraphBatch.symmetric / CSRGraph(symmetric=True): the device-side extraction,
    filter_adj keeps it): the unweighted SAGE backward walks the by-target CSR -- the by-source one is never built -- and
    gives the same dX up to the summation order."""
    N, F = 3000, 128
    half = rand_edges(N, 12000, seed=8, hub=4)
    half = half[:, half[0] != half[1]]
    ei = torch.cat([half, half.flip(0)], dim=1).to(dev)
    x = torch.randn(N, F, generator=torch.Generator().manual_seed(1)).to(dev)
    go = torch.randn(N, 64, generator=torch.Generator().manual_seed(2)).to(dev)
    conv = npi.SAGEConv(F, 64).to(dev)
    outs = []
    for mark in (False, True):
        g = npi.CSRGraph(ei.clone(), N, symmetric=mark)
        xg = x.clone().requires_grad_(True)
        out = conv(xg, g)
        out.backward(go)
        outs.append((out.detach(), xg.grad.clone(), g._by_src is None))
    assert outs[0][2] is False and outs[1][2] is True
    assert torch.equal(outs[0][0], outs[1][0])
    assert torch.allclose(outs[0][1], outs[1][1], atol=1e-5, rtol=1e-5)


@pytest.mark.parametrize("M", [50, 128 * 3 + 17, 4096, 20000 + 9])
@pytest.mark.parametrize("K,N", [(178, 128), (65, 256)])
def test_zero_padded_operand_takes_the_matrix_core_kernels_with_the_same_result(dev, M, K, N):
    """NPI_GEMM_A_ZERO_PADDED: A stored 128-aligned with zero pad columns, W / dW with K rows: fwd and dW equal the fp64
    product at f32 accuracy (and so the unpadded call), for row counts on every path (guarded only, strips, dW slabs)."""
    g = torch.Generator().manual_seed(M + K)
    a = torch.randn(M, K, generator=g)
    w = torch.randn(K, N, generator=g) * 0.1
    b = torch.randn(N, generator=g)
    dc = torch.randn(M, N, generator=g)
    Kp = (K + 127) // 128 * 128
    ap = torch.zeros(M, Kp)
    ap[:, :K] = a
    apd, wd, bd, dcd = ap.to(dev), w.to(dev), b.to(dev), dc.to(dev)
    out = NF.linear_fwd(apd, wd, bd).cpu()
    ref = (a.double() @ w.double() + b.double())
    assert out.shape == (M, N)
    assert (out.double() - ref).abs().max() <= 2e-6 * max(1.0, float(ref.abs().max())) * (K ** 0.5)
    plain = NF.linear_fwd(a.to(dev), wd, bd).cpu()
    torch.testing.assert_close(out, plain, atol=1e-5, rtol=1e-5)
    dw, db = NF.linear_bwd_weight(apd, dcd, k_valid=K)
    dw_ref = a.double().t() @ dc.double()
    assert dw.shape == (K, N)
    assert (dw.cpu().double() - dw_ref).abs().max() <= 3e-6 * float(dw_ref.abs().max()) + 1e-5
    torch.testing.assert_close(db.cpu().double(), dc.double().sum(0), atol=1e-3, rtol=1e-5)
    # a width that is not K rounded up to 128 is refused, not guessed
    with pytest.raises(ValueError):
        NF.linear_fwd(torch.zeros(M, Kp + 128, device=dev), wd, bd)


def test_sage_conv_on_zero_padded_features_equals_the_plain_layer(dev):
    """features that are a view of a 128-aligned buffer with zero pad columns (what InteractionGraph.batch returns): the
    layer runs on the buffer; output, dW and db equal the layer on a contiguous copy"""
    g = torch.Generator().manual_seed(5)
    N, E, Fi, Fo = 6000, 40000, 178, 128
    ei = rand_edges(N, E, 11)
    ei = torch.cat([ei, ei.flip(0)], 1).to(dev)
    x = torch.randn(N, Fi, generator=g)
    full = torch.zeros(N, 256)
    full[:, :Fi] = x
    full = full.to(dev)
    xv = npi.GraphBatch(full[:, :Fi], ei, pad_base=full)
    go = torch.randn(N, Fo, generator=g).to(dev)
    res = []
    for inp in (xv, x.to(dev)):
        torch.manual_seed(0)
        conv = npi.SAGEConv(Fi, Fo).to(dev)
        out = conv(inp).x if isinstance(inp, npi.GraphBatch) else conv(inp, ei)
        out.backward(go)
        res.append((out.detach().cpu(), conv.weight.grad.cpu(), conv.bias.grad.cpu()))
    for p, q in zip(*res):                                  # dW sums 6,000 products per element: different summation orders
        torch.testing.assert_close(p, q, atol=2e-6 * float(q.abs().max()) + 1e-6, rtol=1e-5)
    assert res[0][1].shape == (Fi, Fo)


@pytest.mark.parametrize("Fi,padded", [(128, False), (178, True)])
def test_sage_conv_relu_in_the_epilogue_equals_relu_after_the_layer(dev, Fi, padded):
    """SAGEConv(..., relu=True) == F.relu(SAGEConv(...)): output bit-identical (the same GEMM, the clamp in its epilogue),
    gradients equal (threshold_backward on the saved output), on plain and on zero-padded features"""
    g = torch.Generator().manual_seed(3)
    N, E, Fo = 5000, 30000, 128
    ei = rand_edges(N, E, 21)
    ei = torch.cat([ei, ei.flip(0)], 1).to(dev)
    x = torch.randn(N, Fi, generator=g)
    if padded:
        full = torch.zeros(N, 256)
        full[:, :Fi] = x
        full = full.to(dev)
        xd = full[:, :Fi]
        gbd = npi.GraphBatch(xd, ei, pad_base=full)
    else:
        xd = x.to(dev).requires_grad_(True)
    go = torch.randn(N, Fo, generator=g).to(dev)
    res = []
    for fused in (True, False):
        torch.manual_seed(0)
        conv = npi.SAGEConv(Fi, Fo).to(dev)
        if not padded:
            xd.grad = None
        if padded:
            out = conv(gbd, relu=True).x if fused else torch.relu(conv(gbd).x)
        else:
            out = conv(xd, ei, relu=True) if fused else torch.relu(conv(xd, ei))
        out.backward(go)
        res.append((out.detach().clone(), conv.weight.grad.clone(), conv.bias.grad.clone(), None if padded else xd.grad.clone()))
    assert torch.equal(res[0][0], res[1][0])
    assert bool((res[0][0] == 0).any()) and bool((res[0][0] > 0).any())
    for p, q in zip(res[0][1:], res[1][1:]):
        if p is not None:
            torch.testing.assert_close(p, q, atol=1e-6 * float(q.abs().max()) + 1e-7, rtol=1e-6)


def test_split_gemm_reads_nothing_past_the_end_of_its_operand(dev):
    """The persistent split GEMM's producers run a few k-steps past the end of their tile walk and drop the data; those loads
    must stay inside A.  A here fills its own allocator segment to the last byte (a multiple of 2 MiB, allocated after the
    cache was emptied) and has a row-tile count that is not a multiple of the 8 XCD slots -- the shape in which a walk that
    parked on a slot of the rounded-up grid read up to 7 x 128 rows past the end (a GPU memory fault when nothing is mapped
    there: it took the whole test process down in a full-suite run)."""
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    M, K, N = 512 * 301, 1024, 128                          # 4 KiB rows: 301 x 2 MiB; 1,204 row tiles = 8 x 150 + 4
    a = torch.empty((M, K), device=dev).normal_()
    w = torch.randn(K, N, device=dev) * 0.03
    out = NF.linear_fwd(a, w, None)
    torch.cuda.synchronize()
    rows = torch.tensor([0, 1, M // 2, M - 129, M - 128, M - 1], device=dev)
    ref = a[rows].double() @ w.double()
    assert float((out[rows].double() - ref).abs().max()) < 1e-4
    del a, out
    torch.cuda.empty_cache()


@pytest.mark.parametrize("M,F", [(1000, 128), (4097, 178), (3, 1), (20000, 256)])
def test_relu_backward_kernel_equals_threshold_backward(dev, M, F):
    """npi_relu_backward (the mask of SAGEConv(..., relu=True) in the layer's own backward) == autograd's op, bit for bit,
    on contiguous rows and on a view with a row pitch."""
    g = torch.Generator().manual_seed(M + F)
    dy = torch.randn(M, F, generator=g).to(dev)
    y = torch.relu(torch.randn(M, F, generator=g)).to(dev)
    assert torch.equal(NF.relu_backward(dy, y), torch.ops.aten.threshold_backward(dy, y, 0))
    wide = torch.randn(M, F + 5, generator=g).to(dev)
    assert torch.equal(NF.relu_backward(wide[:, :F], y), torch.ops.aten.threshold_backward(wide[:, :F], y, 0))
    # through the layer: relu=True == F.relu(conv(x)), output and gradients
    if F >= 64 and M >= 1000:
        ei = rand_edges(M, 5 * M, seed=3).to(dev)
        W = (torch.randn(F, 64, generator=g) / F ** 0.5).to(dev)
        b = torch.randn(64, generator=g).to(dev)
        x = torch.randn(M, F, generator=g).to(dev)
        go = torch.randn(M, 64, generator=g).to(dev)
        res = []
        for fused in (True, False):
            xd, Wd, bd = (t.clone().requires_grad_(True) for t in (x, W, b))
            out = npi.sage_conv(xd, ei, Wd, bd, relu=fused)
            out = out if fused else torch.relu(out)
            out.backward(go)
            res.append((out.detach(), xd.grad, Wd.grad, bd.grad))
        assert torch.equal(res[0][0], res[1][0])
        for a, c in zip(res[0][1:], res[1][1:]):
            assert torch.allclose(a, c, rtol=1e-5, atol=1e-6 * float(c.abs().max()))


@pytest.mark.parametrize("M,K,N", [(128, 128, 32), (128 * 3 + 17, 128, 64), (1000, 256, 256), (4096 + 5, 384, 256), (70000, 256, 128)])
def test_bwd_data_with_rank2_epilogue_equals_the_product_plus_the_outer_products(dev, M, K, N):
    """npi_linear_bwd_data_rank2: dC W^T + r0 (x) c0 + r1 (x) c1 with the rank-2 term added in the split kernel's store epilogue
    (GATConv's attention terms of dX) against fp64, on one tile, ragged last row tiles (overlapping tiles store the same
    bits twice), both tile widths and many tiles per workgroup; shapes the kernel does not cover completely are refused."""
    g = torch.Generator().manual_seed(M + K + N)
    dc = torch.randn(M, N, generator=g)
    w = torch.randn(K, N, generator=g) * 0.1
    r0, r1 = torch.randn(M, generator=g), torch.randn(M, generator=g)
    c0, c1 = torch.randn(K, generator=g), torch.randn(K, generator=g)
    dcd, wd = dc.to(dev), w.to(dev)
    assert NF.linear_bwd_data_rank2_ok(dcd, wd)
    out = NF.linear_bwd_data_rank2(dcd, wd, r0.to(dev), r1.to(dev), c0.to(dev), c1.to(dev)).cpu()
    ref = dc.double() @ w.double().t() + torch.outer(r0.double(), c0.double()) + torch.outer(r1.double(), c1.double())
    assert out.shape == (M, K)
    assert (out.double() - ref).abs().max() <= 2e-6 * max(1.0, float(ref.abs().max())) * (N ** 0.5)
    # the plain product is untouched by the new template parameter, and adding the outer products afterwards agrees
    plain = NF.linear_bwd_data(dcd, wd).cpu()
    torch.testing.assert_close(out, plain + torch.outer(r0, c0) + torch.outer(r1, c1), atol=1e-5, rtol=1e-5)
    again = NF.linear_bwd_data_rank2(dcd, wd, r0.to(dev), r1.to(dev), c0.to(dev), c1.to(dev)).cpu()
    assert torch.equal(out, again)


def test_rank2_epilogue_refuses_what_the_split_kernel_does_not_cover(dev):
    for M, K, N in ((100, 128, 32), (512, 178, 64), (512, 128, 30), (512, 64, 32)):
        dc, w = torch.zeros(M, N, device=dev), torch.zeros(K, N, device=dev)
        assert not NF.linear_bwd_data_rank2_ok(dc, w)
        if N % 4 == 0:
            with pytest.raises(npi.NpiError):
                NF.linear_bwd_data_rank2(dc, w, torch.zeros(M, device=dev), torch.zeros(M, device=dev),
                                         torch.zeros(K, device=dev), torch.zeros(K, device=dev))


@pytest.mark.parametrize("M,F", [(1, 1), (7, 3), (1000, 128), (5001, 178), (3000, 300)])
def test_l2_normalize_rows_and_its_backward_match_torch(dev, M, F):
    """functional.l2_normalize = torch.nn.functional.normalize(p=2, dim=-1), forward and backward, incl. all-zero rows (the
    clamp: y = 0, dx = dy / eps) and tiny rows; bitwise reproducible.  SAGEConv(normalize=True) ends with it."""
    g = torch.Generator().manual_seed(M * 31 + F)
    x = torch.randn(M, F, generator=g)
    if M > 5:
        x[3] = 0.0                                     # an all-zero row: the clamped branch
        x[4] *= 1e-20
    go = torch.randn(M, F, generator=g)
    xr = x.clone().double().requires_grad_(True)
    ref = torch.nn.functional.normalize(xr, p=2.0, dim=-1)
    ref.backward(go.double())
    xd = x.to(dev).requires_grad_(True)
    out = NF.l2_normalize(xd)
    out.backward(go.to(dev))
    assert torch.allclose(out.detach().cpu().double(), ref.detach(), atol=1e-6, rtol=1e-5)
    scale = float(xr.grad.abs().max())
    assert float((xd.grad.cpu().double() - xr.grad).abs().max()) <= 1e-5 * scale + 1e-6
    xd2 = x.to(dev).requires_grad_(True)
    out2 = NF.l2_normalize(xd2)
    out2.backward(go.to(dev))
    assert torch.equal(out, out2) and torch.equal(xd.grad, xd2.grad)
    # a view with a row pitch (the unaligned lanes)
    wide = torch.zeros(M, F + 3, device=dev)
    wide[:, 1:F + 1] = x.to(dev)
    assert torch.allclose(NF.l2_normalize(wide[:, 1:F + 1]), out.detach(), atol=1e-6, rtol=1e-6)


def test_sage_conv_normalize_true_matches_the_oracle(dev):
    N, E, Fi, Fo = 2000, 15000, 64, 32
    ei = rand_edges(N, E, seed=5, hub=3)
    g = torch.Generator().manual_seed(6)
    x, go = torch.randn(N, Fi, generator=g), torch.randn(N, Fo, generator=g)
    conv = npi.SAGEConv(Fi, Fo, normalize=True).to(dev)
    xd = x.to(dev).requires_grad_(True)
    out = conv(xd, ei.to(dev))
    out.backward(go.to(dev))
    xr = x.clone().requires_grad_(True)
    Wr, br = conv.weight.detach().cpu().clone().requires_grad_(True), conv.bias.detach().cpu().clone().requires_grad_(True)
    ref = torch.nn.functional.normalize(R.sage_conv(xr, ei, Wr, br), p=2.0, dim=-1)
    ref.backward(go)
    assert torch.allclose(out.detach().cpu(), ref.detach(), atol=1e-5, rtol=1e-4)
    assert torch.allclose(xd.grad.cpu(), xr.grad, atol=1e-5, rtol=1e-3)
    x6, W6, b6 = (t.double().clone().requires_grad_(True) for t in (x, Wr.detach(), br.detach()))
    torch.nn.functional.normalize(R.sage_conv(x6, ei, W6, b6), p=2.0, dim=-1).backward(go.double())
    assert rel_max(conv.weight.grad, W6.grad) <= GRAD_REL and rel_max(conv.bias.grad, b6.grad) <= GRAD_REL


@pytest.mark.parametrize("N,E,Fi,Fo,weighted", [(2000, 15000, 64, 32, False), (3000, 40000, 128, 128, False), (900, 7000, 178, 128, True)])
def test_sage_conv_concat_true_matches_the_oracle(dev, N, E, Fi, Fo, weighted):
    """``SAGEConv(concat=True)`` (PyG 1.4.2): no self loop is added -- existing (i, i) columns stay ordinary
    entries, a node without an in-edge aggregates to zero -- and ``[x_i | mean_j x_j] @ W[2 F, Fo] + b``.  Parity unpinned (the
    reference constructs concat=False): the oracle restates the published forward, gradients by autograd."""
    ei = rand_edges(N, E, seed=11, hub=3)
    ei[:, :3] = torch.tensor([[5, 9, 9], [5, 9, 9]])                         # explicit self loops, one of them twice
    ei[1][ei[1] == 7] = 8                                                    # node 7 has no in-edge
    g = torch.Generator().manual_seed(12)
    x, go = torch.randn(N, Fi, generator=g), torch.randn(N, Fo, generator=g)
    ew = torch.rand(ei.size(1), generator=g) * 2.0 if weighted else None
    conv = npi.SAGEConv(Fi, Fo, concat=True).to(dev)
    assert tuple(conv.weight.shape) == (2 * Fi, Fo)
    xd = x.to(dev).requires_grad_(True)
    out = conv(xd, ei.to(dev), edge_weight=None if ew is None else ew.to(dev))
    out.backward(go.to(dev))
    x6, W6, b6 = (t.detach().cpu().double().clone().requires_grad_(True) for t in (x, conv.weight, conv.bias))
    ref = R.sage_conv_concat(x6, ei, W6, b6, edge_weight=None if ew is None else ew.double())
    ref.backward(go.double())
    assert torch.allclose(out.detach().cpu().double(), ref.detach(), atol=ATOL, rtol=RTOL)
    assert torch.allclose(xd.grad.cpu().double(), x6.grad, atol=ATOL, rtol=RTOL)
    assert rel_max(conv.weight.grad, W6.grad) <= GRAD_REL and rel_max(conv.bias.grad, b6.grad) <= GRAD_REL
    assert float(out.detach()[7].cpu().double().sub(x6.detach()[7] @ W6.detach()[:Fi] + b6.detach()).abs().max()) <= 1e-5   # no in-edge: own features only
    # a prebuilt graph must be the edge list as it is
    with pytest.raises(ValueError):
        conv(xd, npi.CSRGraph(ei.to(dev), N))
    gg = npi.CSRGraph(ei.to(dev), N, self_loops=False, keep_equal=True)
    assert int(gg.by_dst.rowptr[-1]) == ei.size(1)                           # nothing dropped, nothing appended
    if not weighted:
        assert torch.equal(conv(x.to(dev), gg).detach(), out.detach())


@pytest.mark.parametrize("item_entries", [64, 256])
@pytest.mark.parametrize("F", [256, 64, 20, 300])
def test_long_chains_of_partials_add_up_exactly(dev, item_entries, F):
    """Rows cut over MANY items (a hub row: 7,800 partials at C5) through the carry + fix-up path.  Integer-valued features make
    every order of addition exact in f32, so the result must EQUAL the reference bit for bit: rows starting / ending exactly on
    multiples of 64 items, rows one entry short of that, an empty row in between, plus mean, per-entry weights and a repeat
    run.  Both item sizes; one-chunk, narrow (group kernel) and two-chunk rows.  (Written for a variant that pre-summed spans of
    64 partials in an extra launch -- measured slower everywhere and dropped; the test stays.)"""
    span = 64 * item_entries                                           # entries of 64 items
    lens = [5, span - 5,                                               # -> the next row starts exactly at entry `span`
            3 * span,                                                  # starts on a boundary, ends on one
            7, 2 * span + 11, 0, span - 1, 20 * span + 123, 1, 2 * span - 7, 5 * span]
    g = torch.Generator().manual_seed(F + item_entries)
    small = torch.randint(0, 40, (30000,), generator=g).tolist()
    lens = lens + small
    n_rows = len(lens)
    nnz = sum(lens)
    n_cols = 5000
    key = torch.repeat_interleave(torch.arange(n_rows), torch.tensor(lens))
    val = torch.randint(0, n_cols, (nnz,), generator=g)
    x = torch.randint(-3, 4, (n_cols, F), generator=g).float()
    w = torch.randint(-2, 3, (nnz,), generator=g).float()
    side = NG.build_side(key.to(dev), val.to(dev), n_rows, n_cols, self_loops=False, drop_equal=False, item=item_entries)
    assert side.item == item_entries
    xd = x.to(dev)
    ref = torch.zeros(n_rows, F).index_add_(0, key, x[val])           # (small integers: exact in f32 in any order)
    out = NF.segsum(None, side, xd)
    assert torch.equal(out.cpu(), ref)
    assert torch.equal(NF.segsum(None, side, xd), out)
    # per-entry weights (entry order = edge order here: the keys are sorted and the build is stable)
    we = torch.empty(side.nnz_max, device=dev)
    we[:nnz] = w.to(dev)[side.eid[:nnz].long()]
    refw = torch.zeros(n_rows, F).index_add_(0, key, x[val] * w.view(-1, 1))
    assert torch.equal(NF.segsum(None, side, xd, w=we).cpu(), refw)
    # mean: one division per row after the exact sum
    cnt = torch.tensor(lens, dtype=torch.float64).clamp(min=1).view(-1, 1)
    got = NF.segsum(None, side, xd, mean=True).cpu().double()
    assert torch.allclose(got, ref.double() / cnt, rtol=1e-6, atol=0)


@pytest.mark.parametrize("item_entries", [64, 256])
def test_in_launch_chain_resolution_under_load_and_reuse(dev, item_entries):
    """Rows cut by a workgroup boundary are summed INSIDE the aggregation launch by whichever workgroup delivers a row's last
    partial (write-through stores, agent-scope arrival counters, one acquire: segsum.hip).  What would break such a hand-off
    shows only under reuse and uneven load (cdna_hip_programming.md Guideline 16, pitfalls 3 / 10): so the SAME scratch buffer
    serves 60 launches that alternate between two inputs -- a partial read stale from the previous launch belongs to the other
    input and changes the (integer-valued, hence exact) sums -- while a second stream keeps the chip busy with another
    aggregation of changing size, and the sums are compared bit for bit every time.  Long chains (two-level sums), chains that
    end on span boundaries, short chains and thousands of two-partial rows."""
    span = 4 * 64 * item_entries                                       # entries of 64 workgroups (one span of the two-level sum)
    lens = [3, span + 17, 2 * span, 5, 64 * item_entries - 1, 7 * span + 3, 1, 4 * item_entries, 4 * item_entries + 1, 9]
    g = torch.Generator().manual_seed(item_entries)
    lens = lens + torch.randint(0, 3 * item_entries, (4000,), generator=g).tolist()
    n_rows, nnz, n_cols, F = len(lens), sum(lens), 3000, 256
    key = torch.repeat_interleave(torch.arange(n_rows), torch.tensor(lens))
    val = torch.randint(0, n_cols, (nnz,), generator=g)
    side = NG.build_side(key.to(dev), val.to(dev), n_rows, n_cols, self_loops=False, drop_equal=False, item=item_entries)
    xs = [torch.randint(-3, 4, (n_cols, F), generator=g).float() for _ in range(2)]
    refs = [torch.zeros(n_rows, F).index_add_(0, key, x[val]).to(dev) for x in xs]
    xd = [x.to(dev) for x in xs]
    # the load on the other stream: aggregations over another side (its own scratch), sizes changing from launch to launch
    lk = torch.randint(0, 2000, (300_000,), generator=g)
    lv = torch.randint(0, n_cols, (300_000,), generator=g)
    noise = [NG.build_side(lk[:n].to(dev), lv[:n].to(dev), 2000, n_cols, self_loops=False, drop_equal=False)
             for n in (300_000, 40_000, 150_000)]
    other = torch.cuda.Stream(device=dev)
    out = torch.empty(n_rows, F, device=dev)
    bad = 0
    for it in range(60):
        with torch.cuda.stream(other):
            NF.segsum(None, noise[it % 3], xd[it % 2])
        k = (it * 7 + it // 3) % 2
        NF.segsum(None, side, xd[k], out=out)
        bad += int((out != refs[k]).any())
    torch.cuda.synchronize()
    assert bad == 0
    # the scratch is left reusable: every arrival counter is back at zero
    n_items = side.n_items
    n_wg = -(-n_items // 4)
    counters = side.carry(F)[: n_wg + 2 * (-(-n_wg // 64))].view(torch.int32)
    assert int(counters.abs().sum()) == 0


def test_a_csr_carries_the_item_size_it_was_built_with(dev):
    """VERDICT r3 item 4 / ADVICE r3 (medium): the item size used to be re-derived at every LAUNCH from a process-wide
    threshold, so a CSR built before that threshold moved and aggregated after it was walked with the wrong geometry.  Now
    the side carries it and NOTHING that could move exists: the same edge list built with 64-entry and with
    256-entry items (``CSRGraph(item=)``; the hint for this capacity names one of the two) -- every consumer of either side, SAGE
    aggregation (plain, weighted, bf16), the GAT forward and fused backward, the whole layers, matches the oracle."""
    N, E, F = 3000, 1_200_000, 64                     # capacity 2.4M entries: between the two thresholds used below
    ei = rand_edges(N, E, seed=7)
    x = torch.randn(N, F, generator=torch.Generator().manual_seed(1))
    go = torch.randn(N, F, generator=torch.Generator().manual_seed(2))
    if True:
        for item in (64, 256):
            g = npi.CSRGraph(ei.to(dev), N, item=item)
            assert g.by_dst.item == item and g.by_src.item == item and NG.item_hint(g.by_dst.nnz_max) in (64, 256)
            sage = npi.SAGEConv(F, F).to(dev)
            gat = npi.GATConv(F, F).to(dev)
            xd = x.to(dev).requires_grad_(True)
            out = sage(xd, g)
            out.backward(go.to(dev))
            r_out, r_dx, r_dw, r_db = R.sage_layer_fwd_bwd(x, ei, sage.weight.detach().cpu(), sage.bias.detach().cpu(), go)
            assert float((out.detach().cpu() - r_out).abs().max()) <= 1e-4
            assert float((xd.grad.cpu() - r_dx).abs().max()) <= 1e-4
            dw64 = R.sage_layer_fwd_bwd(x.double(), ei, sage.weight.detach().cpu().double(), sage.bias.detach().cpu().double(), go.double())[2]
            assert rel_max(sage.weight.grad, dw64) <= GRAD_REL
            # bf16 storage and per-entry weights walk the same item_row / carry
            agg16 = NF.segsum(g, g.by_dst, x.to(dev).bfloat16(), mean=True).float().cpu()
            agg32 = NF.segsum(g, g.by_dst, x.to(dev), mean=True).cpu()
            assert float((agg16 - agg32).abs().max()) <= 2e-2 * float(agg32.abs().max())
            # GATConv: statistics, forward aggregation on the read-back scores, fused backward, both row sums
            xg = x.to(dev).requires_grad_(True)
            og = gat(xg, g)
            og.backward(go.to(dev))
            xr = x.clone().double().requires_grad_(True)
            ref = R.gat_conv(xr, ei, gat.weight.detach().cpu().double(), gat.att.detach().cpu().double(),
                             gat.bias.detach().cpu().double(), heads=1)
            ref.backward(go.double())
            assert float((og.detach().cpu().double() - ref.detach()).abs().max()) <= 1e-4
            assert float((xg.grad.cpu().double() - xr.grad).abs().max()) <= 1e-4 * max(1.0, float(xr.grad.abs().max()))


@pytest.mark.parametrize("M,K,N", [(5085, 128, 128), (4096, 178, 128), (70_003, 256, 256), (20, 64, 64), (33, 96, 130), (1024, 128, 64)])
@pytest.mark.parametrize("want_bias", [True, False])
def test_bf16_dw_matches_f32_reference(dev, M, K, N, want_bias):
    """bf16 storage: dW = A^T dC and db = colsum(dC) (f32 slabs on the matrix cores, then ONE finishing launch that adds the
    slabs, the < 32 trailing nodes and db and rounds to bf16 -- dw_finish_kernel<bf16>; fewer than 32 nodes: the guarded path)
    against an f32 product of the same bf16-rounded inputs; bitwise reproducible."""
    g = torch.Generator().manual_seed(M + K + N)
    A = torch.randn(M, K, generator=g).to(torch.bfloat16)
    dC = torch.randn(M, N, generator=g).to(torch.bfloat16)
    ref_w = A.float().t() @ dC.float()
    ref_b = dC.float().sum(0)
    dw, db = NF.linear_bwd_weight(A.to(dev), dC.to(dev), want_bias)
    assert dw.dtype == torch.bfloat16 and dw.shape == (K, N) and (db is None) == (not want_bias)
    scale = float(ref_w.abs().max())
    assert float((dw.float().cpu() - ref_w).abs().max()) <= 2.0 ** -7 * scale            # one bf16 rounding of the result
    if want_bias:
        assert float((db.float().cpu() - ref_b).abs().max()) <= 2.0 ** -7 * float(ref_b.abs().max())
    dw2, db2 = NF.linear_bwd_weight(A.to(dev), dC.to(dev), want_bias)
    assert torch.equal(dw, dw2) and (db is None or torch.equal(db, db2))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("M,K,N", [(1000, 256, 256), (5085, 128, 128), (300, 64, 192), (130, 160, 128)])
def test_prepared_weight_copies_give_the_same_bits(dev, dtype, M, K, N):
    """npi_linear_prepare writes both re-laid copies of W in one launch; a GEMM told NPI_GEMM_WORKSPACE_PREPARED reads them
    instead of preparing its own: forward and bwd_data bit-equal to the plain calls, also on row blocks (the sharded layers'
    light / hub split), f32 and bf16 storage."""
    g = torch.Generator().manual_seed(M + K + N)
    A = torch.randn(M, K, generator=g).to(dtype).to(dev)
    W = (torch.randn(K, N, generator=g) / K ** 0.5).to(dtype).to(dev)
    b = torch.randn(N, generator=g).to(dtype).to(dev)
    dC = torch.randn(M, N, generator=g).to(dtype).to(dev)
    rs = torch.rand(M, generator=g).to(dev)
    wsf, wsb = NF.prepare_weight(W)
    assert wsf is not None and wsb is not None
    ref_f, ref_b = NF.linear_fwd(A, W, b, relu=True), NF.linear_bwd_data(dC, W, rs)
    assert torch.equal(NF.linear_fwd(A, W, b, relu=True, ws=wsf), ref_f)
    assert torch.equal(NF.linear_bwd_data(dC, W, rs, ws=wsb), ref_b)
    only_f, none = NF.prepare_weight(W, backward=False)
    assert none is None and torch.equal(NF.linear_fwd(A, W, b, relu=True, ws=only_f), ref_f)
    assert NF.prepare_weight(W[:, : N - 8].contiguous()) == (None, None)             # a width the matrix-core kernels do not take
    h = M // 3
    if h < 128:                          # a block of < 128 rows takes the guarded exact-f32 kernel: other arithmetic either way
        return
    out = torch.empty_like(ref_f)
    NF.linear_fwd(A[:h], W, b, relu=True, out=out[:h], ws=wsf)
    NF.linear_fwd(A[h:], W, b, relu=True, out=out[h:], ws=wsf)
    assert torch.equal(out, ref_f)
    da = torch.empty_like(ref_b)
    NF.linear_bwd_data(dC[h:], W, rs[h:], out=da[h:], ws=wsb)
    NF.linear_bwd_data(dC[:h], W, rs[:h], out=da[:h], ws=wsb)
    assert torch.equal(da, ref_b)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("N,K,Fo", [(5000, 178, 128), (700, 90, 64), (3000, 200, 256), (1500, 300, 128), (2000, 65, 128), (120, 178, 128),
                                    (4000, 100, 160)])
def test_sage_layer_with_odd_input_width_matches_the_oracle(dev, dtype, N, K, Fo):
    """SAGEConv fwd + bwd at input widths that are not a multiple of 128: from 128 rows on and when the padding costs < 1.5 x the
    aggregate is kept 128-aligned with zero pad columns (f32: NPI_GEMM_A_ZERO_PADDED; bf16: W padded with zero rows, dAgg / dW
    cut back) -- 178 -> 256, 90 / 100 -> 128, 200 -> 256, 300 -> 384 padded; 65 and the 120-row graph not -- against the CPU
    oracle on the same (bf16-rounded) inputs."""
    g = torch.Generator().manual_seed(N + K)
    E = 8 * N
    ei = torch.randint(0, N, (2, E), generator=g)
    x = torch.randn(N, K, generator=g)
    W = torch.randn(K, Fo, generator=g) / K ** 0.5
    b = torch.randn(Fo, generator=g) * 0.1
    go = torch.randn(N, Fo, generator=g)
    if dtype == torch.bfloat16:
        x, W, b, go = (t.to(dtype).float() for t in (x, W, b, go))
    xr, Wr, br = (t.clone().requires_grad_(True) for t in (x, W, b))
    ref = R.sage_conv(xr, ei, Wr, br)
    ref.backward(go)
    xd, Wd, bd = (t.to(dev).to(dtype).requires_grad_(True) for t in (x, W, b))
    out = npi.sage_conv(xd, npi.CSRGraph(ei.to(dev), N), Wd, bd)
    out.backward(go.to(dev).to(dtype))
    tol = 2e-2 if dtype == torch.bfloat16 else 1e-4
    for name, got, want in (("out", out, ref), ("dx", xd.grad, xr.grad), ("dW", Wd.grad, Wr.grad), ("db", bd.grad, br.grad)):
        assert got.shape == want.shape and got.dtype == dtype, name
        err = float((got.detach().float().cpu() - want.detach()).abs().max() / (want.detach().abs().max() + 1e-12))
        assert err <= tol, (name, err)
