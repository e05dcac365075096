"""Randomised shapes against the oracle: CSR build bit-exact (row pointers, column / edge ids in reference
order), aggregation in both orientations against an fp64 oracle, and the shapes that sit on the internal
boundaries -- rows ending exactly at a 64- / 256-entry item boundary, entry counts that are exact multiples
of the item size, all-empty graphs, one giant row."""
import numpy as np
import pytest
import torch

import npi_gnn_amd as npi
from npi_gnn_amd import functional as NF
from oracle import ref_conv as R
from test_gpu_parity import csr_reference

pytestmark = pytest.mark.gpu


def _check(dev, ei, N, F, loops, mean, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(N, F, generator=g)
    graph = npi.CSRGraph(ei.to(dev), N, self_loops=loops)
    for side, key, val in ((graph.by_dst, ei[1], ei[0]), (graph.by_src, ei[0], ei[1])):
        rowptr, col, eid, rowidx = csr_reference(key, val, N, loops)
        nnz = int(rowptr[-1])
        assert np.array_equal(side.rowptr.cpu().numpy(), rowptr)
        assert np.array_equal(side.col.cpu().numpy()[:nnz], col)
        assert np.array_equal(side.eid.cpu().numpy()[:nnz], eid)
        out = NF.segsum(graph, side, x.to(dev), mean=mean).cpu()
        ref = torch.zeros(N, F, dtype=torch.float64)
        if nnz:
            ref.index_add_(0, torch.from_numpy(rowidx).long(), x.double()[torch.from_numpy(col).long()])
        if mean:
            ref = ref / torch.from_numpy(np.diff(rowptr)).clamp(min=1).double().view(-1, 1)
        scale = max(1.0, float(ref.abs().max()))
        assert float((out.double() - ref).abs().max()) <= 2e-5 * scale, (N, ei.size(1), F, loops, mean)


@pytest.mark.parametrize("seed", range(24))
def test_random_graphs(dev, seed):
    rng = np.random.default_rng(seed)
    N = int(rng.integers(1, 3000))
    E = int(rng.integers(0, 20000))
    F = int(rng.choice([1, 7, 64, 128, 178, 256, 300]))
    ei = torch.from_numpy(rng.integers(0, N, size=(2, E)))
    if E and rng.random() < 0.5:
        ei[1, : E // 2] = int(rng.integers(0, N))            # a hub target
    if E and rng.random() < 0.3:
        ei[0, E // 2:] = int(rng.integers(0, N))             # a hub source
    _check(dev, ei, N, F, bool(rng.random() < 0.7), bool(rng.random() < 0.5), seed)


@pytest.mark.parametrize("deg,N,F", [(63, 40, 128), (64, 40, 256), (255, 9, 64), (256, 9, 256), (128, 33, 178), (1, 500, 128),
                                     (255, 4200, 64), (256, 4200, 32)])        # the last two: > 2^20 entries, 256-entry items
def test_rows_on_item_boundaries(dev, deg, N, F):
    """every row has `deg` distinct in-neighbours: with the self loop (or without) rows end exactly on item boundaries"""
    src = torch.cat([(torch.arange(deg) + r + 1) % max(N, deg + 2) for r in range(N)])
    dst = torch.arange(N).repeat_interleave(deg)
    M = max(N, deg + 2)
    ei = torch.stack([src, dst])
    for loops in (True, False):
        _check(dev, ei, M, F, loops, True, deg)


def test_degenerate_graphs(dev):
    _check(dev, torch.zeros((2, 0), dtype=torch.long), 1, 64, True, True, 0)            # one node, no edge
    _check(dev, torch.zeros((2, 0), dtype=torch.long), 300, 256, True, False, 0)        # only self loops
    _check(dev, torch.tensor([[0, 1, 2], [0, 1, 2]]), 3, 128, True, True, 0)            # only explicit self loops
    ei = torch.stack([torch.arange(1, 5000), torch.zeros(4999, dtype=torch.long)])      # one giant row
    _check(dev, ei, 5000, 128, True, True, 1)
    _check(dev, ei.flip(0), 5000, 256, False, False, 2)                                 # one giant source row, no loops


# ---- between-layer steps: TopKPooling + filter_adj + readout on random batches ---------------------------------
@pytest.mark.parametrize("seed", range(16))
def test_random_pooling_batches(dev, seed):
    """graph sizes 0..2,500 (so both LDS sort capacities and empty graphs occur), ratios incl. the reference's 0.5,
    duplicated rows (tied scores: lower index wins), widths that are / are not multiples of 4; indices bit-exact."""
    from npi_gnn_amd import pool as NP
    rng = np.random.default_rng(1000 + seed)
    g = torch.Generator().manual_seed(seed)
    n_graphs = int(rng.integers(1, 40))
    sizes = rng.choice([0, 1, 2, 3, 7, 64, 65, 300, 1024, 1025, 2500], size=n_graphs,
                       p=[.08, .1, .1, .1, .15, .1, .1, .12, .05, .05, .05])
    if sizes.sum() == 0:
        sizes[0] = 5
    sizes = [int(v) for v in sizes]
    F = int(rng.choice([1, 6, 16, 128, 178]))
    ratio = float(rng.choice([0.5, 0.5, 0.25, 0.8, 1.0]))
    batch = torch.cat([torch.full((n,), b, dtype=torch.long) for b, n in enumerate(sizes)])
    N = batch.numel()
    x = torch.randn(N, F, generator=g)
    if N > 8:                                                # ties: a quarter of the rows are copies of earlier rows
        idx = torch.from_numpy(rng.integers(0, N, size=N // 4))
        x[idx] = x[torch.from_numpy(rng.integers(0, N, size=N // 4))]
    w = torch.randn(1, F, generator=g)
    starts = np.concatenate([[0], np.cumsum(sizes)])
    src, dst = [], []
    for b, n in enumerate(sizes):
        if n:
            e = torch.from_numpy(rng.integers(0, n, size=(2, 2 * n))) + int(starts[b])
            src.append(e[0]); dst.append(e[1])
    ei = torch.stack([torch.cat(src), torch.cat(dst)])
    from npi_gnn_amd.graph import GraphBatch
    gin = GraphBatch(x.to(dev), ei.to(dev), batch.to(dev), n_graphs)
    sel = NP._select(gin, w.to(dev), ratio, False)
    (gx, gsc), ge, gb, gperm, score_gpu = NP._gather(gin.x, sel), sel.edge_index, sel.batch_out, sel.perm64, sel.score
    # scores: float rounding only (tanh and the dot product are not bit-identical across libraries)
    score = torch.tanh((x * w.view(1, -1)).sum(-1) / w.norm(p=2))
    assert torch.allclose(score_gpu.cpu(), score, atol=1e-6)
    # selection: integer work, bit-exact GIVEN the scores -- k = ceil(ratio * n) in float32 like PyG 1.4.2, descending,
    # ties to the lower index, graphs in order; two scores one ulp apart would otherwise make the order a coin toss
    sg = score_gpu.cpu()
    perm = []
    for b_, n in enumerate(sizes):
        k = int(torch.ceil(torch.tensor(ratio, dtype=torch.float32) * torch.tensor(float(n), dtype=torch.float32)))
        o = torch.sort(sg[int(starts[b_]): int(starts[b_]) + n], descending=True, stable=True).indices[:k]
        perm.append(o + int(starts[b_]))
    perm = torch.cat(perm)
    assert torch.equal(gperm.cpu(), perm), (sizes, F, ratio)
    remap = torch.full((N,), -1, dtype=torch.long)
    remap[perm] = torch.arange(perm.numel())
    r_, c_ = remap[ei[0]], remap[ei[1]]
    keep = (r_ >= 0) & (c_ >= 0)
    eo, bo = torch.stack([r_[keep], c_[keep]]), batch[perm]
    xo, sc = x[perm] * sg[perm].view(-1, 1), sg[perm]
    assert torch.equal(gb.cpu(), bo) and torch.equal(ge.cpu(), eo)
    assert torch.allclose(gx.cpu(), xo, atol=1e-6, rtol=1e-6) and torch.equal(gsc.cpu(), sc)
    ro = NP.global_max_mean_pool(gx, gb, n_graphs).cpu()
    ref = R.readout(xo, bo, n_graphs)
    kept = torch.bincount(bo, minlength=n_graphs) > 0
    assert torch.allclose(ro[kept], ref[kept], atol=1e-5, rtol=1e-5)
    assert float(ro[~kept].abs().sum()) == 0.0               # a graph without nodes reads out zeros


def test_gat_random_campaign(dev):
    """tools/fuzz_gat.py, 24 cases (the 60-case campaign: tests/test_slow_campaigns.py, marker `slow`): random graphs (hub rows, empty rows, self loops, duplicates), 1 / 2 / 4 / 8 heads, both
    item sizes (CSRGraph(item=)), fused ReLU on / off -- GATConv
    forward and every gradient against the fp64 oracle (1,050 cases of the same generator ran clean in round 3)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_gat.py"), "24", "11"], capture_output=True, text=True,
                       timeout=600)
    assert p.returncode == 0 and "24 cases ok" in p.stdout, (p.stdout + p.stderr)[-2000:]


def test_sharded_layers_random_campaign(dev):
    """tools/fuzz_dist.py, 8 cases (24: tests/test_slow_campaigns.py): W = 1..8 virtual ranks on this GPU, random bipartite / arbitrary graphs, SAGE / GCN / GAT with
    1-8 heads, against the single-GPU layers (360 cases of the same generator ran clean in round 3, worst error 1.2e-6)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_dist.py"), "8", "7"], capture_output=True, text=True,
                       timeout=900)
    assert p.returncode == 0 and "8 cases ok" in p.stdout, (p.stdout + p.stderr)[-2000:]
