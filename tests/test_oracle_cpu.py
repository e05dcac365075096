"""The oracle against independent dense formulas and fp64 gradcheck (CPU; no GPU needed).
Golden-vector pinning of the oracle against the reference's own artifacts is in test_golden.py."""
import torch

from oracle import ref_conv as R


def _rand_graph(N, E, seed, self_loops=True):
    g = torch.Generator().manual_seed(seed)
    ei = torch.randint(0, N, (2, E), generator=g)
    if not self_loops:
        ei = ei[:, ei[0] != ei[1]]
    return ei


def _dense_adj(ei, N, w=None):
    A = torch.zeros(N, N, dtype=torch.float64)
    for k, (s, d) in enumerate(ei.t().tolist()):
        A[d, s] += 1.0 if w is None else float(w[k])
    return A


def test_sage_matches_dense_mean_with_self_loops():
    N, E, F = 12, 40, 5
    ei = _rand_graph(N, E, 0)
    x = torch.randn(N, F, dtype=torch.float64)
    W = torch.randn(F, 3, dtype=torch.float64)
    b = torch.randn(3, dtype=torch.float64)
    keep = ei[0] != ei[1]
    A = _dense_adj(ei[:, keep], N) + torch.eye(N, dtype=torch.float64)
    ref = (A @ x) / A.sum(1, keepdim=True) @ W + b
    assert torch.allclose(R.sage_conv(x, ei, W, b), ref, atol=1e-12)


def test_direction_is_source_to_target():
    # single directed edge 0 -> 1: node 1 averages {x0, x1}; node 0 keeps x0
    x = torch.tensor([[1.0], [3.0]], dtype=torch.float64)
    ei = torch.tensor([[0], [1]])
    agg = R.sage_aggregate(x, ei)
    assert agg[:, 0].tolist() == [1.0, 2.0]


def test_gcn_matches_dense_symmetric_norm():
    N, E, F = 10, 30, 4
    ei = _rand_graph(N, E, 1, self_loops=False)
    ei = torch.cat([ei, ei.flip(0)], dim=1)      # symmetric, as every reference graph is
    x = torch.randn(N, F, dtype=torch.float64)
    W = torch.randn(F, 6, dtype=torch.float64)
    b = torch.randn(6, dtype=torch.float64)
    A = _dense_adj(ei, N) + torch.eye(N, dtype=torch.float64)
    d = A.sum(0)                                  # out-degree over sources (== in-degree here)
    ref = (d.pow(-0.5).view(-1, 1) * A * d.pow(-0.5).view(1, -1)) @ (x @ W) + b
    assert torch.allclose(R.gcn_conv(x, ei, W, b), ref, atol=1e-12)


def test_gat_rows_are_convex_combinations():
    N, E, F, H, C = 9, 25, 4, 2, 3
    ei = _rand_graph(N, E, 2)
    x = torch.randn(N, F, dtype=torch.float64)
    W = torch.randn(F, H * C, dtype=torch.float64)
    att = torch.zeros(1, H, 2 * C, dtype=torch.float64)     # zero attention => plain mean
    out = R.gat_conv(x, ei, W, att, None, heads=H)
    ref = R.sage_aggregate(x @ W, ei)
    assert torch.allclose(out, ref, atol=1e-12)


def test_gradcheck_fp64():
    N, E, F = 7, 15, 3
    ei = _rand_graph(N, E, 3)
    x = torch.randn(N, F, dtype=torch.float64, requires_grad=True)
    W = torch.randn(F, 2, dtype=torch.float64, requires_grad=True)
    b = torch.randn(2, dtype=torch.float64, requires_grad=True)
    assert torch.autograd.gradcheck(lambda x_, W_, b_: R.sage_conv(x_, ei, W_, b_), (x, W, b))
    assert torch.autograd.gradcheck(lambda x_, W_, b_: R.gcn_conv(x_, ei, W_, b_), (x, W, b))


def test_topk_pool_keeps_ceil_half_per_graph():
    x = torch.randn(7, 4)
    batch = torch.tensor([0, 0, 0, 1, 1, 1, 1])
    ei = torch.tensor([[0, 1, 3, 4, 5], [1, 2, 4, 5, 6]])
    xo, eo, bo, perm, score = R.topk_pool(x, ei, batch, torch.randn(1, 4))
    assert bo.tolist() == [0, 0, 1, 1]            # ceil(1.5)=2, ceil(2)=2
    assert (score[0] >= score[1]) and (score[2] >= score[3])
    assert eo.numel() == 0 or int(eo.max()) < 4


def test_metrics_formula_matches_reference_log_line():
    # result/1223_1/log_0.txt: "result, testing dataset, Accuracy: 0.93495, Precision: 0.91636,
    # Sensitivity: 0.95727, Specificity: 0.91263, MCC: 0.87077"  <- TP 1994 FN 89 TN 1901 FP 182
    got = ["%.5f" % v for v in R.metrics_from_confusion(1994, 89, 1901, 182)]
    assert got == ["0.93495", "0.91636", "0.95727", "0.91263", "0.87077"]


def test_sage_conv_concat_restatement_by_hand_and_gradcheck():
    """oracle.sage_conv_concat (PyG 1.4.2 SAGEConv(concat=True)): no self loop added, existing ones are ordinary messages, an
    isolated target aggregates to zero; weight [2 F, Fo]"""
    x = torch.tensor([[1.0, 2.0], [3.0, 5.0], [7.0, 11.0]], dtype=torch.float64)
    ei = torch.tensor([[0, 1, 1], [1, 1, 0]])                       # 0 -> 1, 1 -> 1 (a self loop), 1 -> 0; node 2 has no in-edge
    W = torch.arange(8, dtype=torch.float64).view(4, 2) / 10
    b = torch.tensor([0.5, -0.5], dtype=torch.float64)
    agg = torch.stack([x[1], (x[0] + x[1]) / 2, torch.zeros(2, dtype=torch.float64)])
    want = torch.cat([x, agg], dim=1) @ W + b
    assert torch.allclose(R.sage_conv_concat(x, ei, W, b), want)
    g = torch.Generator().manual_seed(0)
    ei2 = torch.randint(0, 9, (2, 30), generator=g)
    xs, Ws, bs = (torch.randn(*s, dtype=torch.float64, generator=g, requires_grad=True) for s in ((9, 5), (10, 3), (3,)))
    assert torch.autograd.gradcheck(lambda a, w, c: R.sage_conv_concat(a, ei2, w, c), (xs, Ws, bs))


def test_gat_keep_scale_is_dropout_on_the_normalised_attention_weights():
    """oracle.gat_conv(keep_scale=): PyG 1.4.2 GATConv.message drops attention weights AFTER the softmax
    (``alpha = softmax(alpha, edge_index_i); alpha = F.dropout(alpha, p, training)``): a dropped edge does not renormalise the
    others.  By hand on a 3-node graph with zero attention (alpha = 1 / in-count incl. the loop), and gradcheck."""
    x = torch.tensor([[1.0, 0.0], [0.0, 1.0], [2.0, 2.0]], dtype=torch.float64)
    ei = torch.tensor([[1, 2, 0], [0, 0, 1]])                  # 1 -> 0, 2 -> 0, 0 -> 1; loops (0,0) (1,1) (2,2) are appended
    W = torch.eye(2, dtype=torch.float64)
    att = torch.zeros(1, 1, 4, dtype=torch.float64)
    ks = torch.tensor([[2.0], [0.0], [2.0], [2.0], [0.0], [2.0]], dtype=torch.float64)     # p = 0.5: edges 2->0 and loop (1,1) dropped
    out = R.gat_conv(x, ei, W, att, None, heads=1, keep_scale=ks)
    want = torch.stack([(x[1] + x[0]) * 2 / 3,                 # target 0: alpha = 1/3 each of {1, 2, loop}, 2 -> 0 dropped
                        x[0] * 2 / 2,                          # target 1: alpha = 1/2 each of {0, loop}, the loop dropped
                        x[2] * 2])                             # target 2: its loop only, kept
    assert torch.allclose(out, want)
    g = torch.Generator().manual_seed(1)
    ei2 = _rand_graph(8, 20, 4)
    Ek = int((ei2[0] != ei2[1]).sum())
    ks2 = (torch.rand(Ek + 8, 2, generator=g) < 0.7).double() / 0.7
    xs, Ws, As = (torch.randn(*s, dtype=torch.float64, generator=g, requires_grad=True) for s in ((8, 3), (3, 4), (1, 2, 4)))
    assert torch.autograd.gradcheck(lambda a, w, t: R.gat_conv(a, ei2, w, t, None, heads=2, keep_scale=ks2), (xs, Ws, As))
