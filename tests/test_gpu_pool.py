"""TopKPooling / readout kernels against the oracle, and the WHOLE Net_1
forward on the MI355X against the reference's own logged results (golden fixtures)."""
import os

import pytest
import torch
import torch.nn.functional as F

import npi_gnn_amd as npi
from npi_gnn_amd import graph as NG
from npi_gnn_amd import pool as NP
from oracle import kat, ref_conv as R

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _pool_tuple(gb, w, ratio, padded_edges=False):
    """topk_pool_batch in the shape of PyG's 6-tuple (the tests below index it like the oracle's)"""
    out, perm, score = NP.topk_pool_batch(gb, w, ratio, padded_edges=padded_edges)
    return out.x, out.edge_index, None, out.batch, perm, score


def load(name):
    return torch.load(os.path.join(G, name), map_location="cpu", weights_only=False)


class Net1(torch.nn.Module):
    """The wiring of the reference's Net_1 (src/classes.py:45-82) with this package's modules:
    3 x (SAGEConv -> relu -> TopKPooling(0.5) -> [gmp || gap]) summed -> MLP -> log_softmax."""

    def __init__(self, num_node_features, num_of_classes=2):
        super().__init__()
        self.conv1, self.pool1 = npi.SAGEConv(num_node_features, 128), NP.TopKPooling(128, ratio=0.5)
        self.conv2, self.pool2 = npi.SAGEConv(128, 128), NP.TopKPooling(128, ratio=0.5)
        self.conv3, self.pool3 = npi.SAGEConv(128, 128), NP.TopKPooling(128, ratio=0.5)
        self.lin1 = torch.nn.Linear(256, 128)
        self.lin2 = torch.nn.Linear(128, 64)
        self.lin3 = torch.nn.Linear(64, num_of_classes)
        self.dropout_p = 0.5

    def forward(self, x, edge_index, batch):
        acc = None
        for conv, pool in ((self.conv1, self.pool1), (self.conv2, self.pool2), (self.conv3, self.pool3)):
            x = F.relu(conv(x, edge_index))
            x, edge_index, _, batch, _, _ = pool(x, edge_index, None, batch)
            r = torch.cat([NP.global_max_pool(x, batch), NP.global_mean_pool(x, batch)], dim=1)
            acc = r if acc is None else acc + r
        x = F.relu(self.lin1(acc))
        x = F.dropout(x, p=self.dropout_p, training=self.training)
        x = F.relu(self.lin2(x))
        return F.log_softmax(self.lin3(x), dim=-1)


def test_topk_pool_matches_oracle_bit_exact_indices(dev):
    fx = load("npinter2_small.pt")
    x = torch.relu(fx["conv_out"][0])                       # what pool1 sees
    ei, batch = fx["edge_index"], fx["batch"]
    w = fx["state_dict"]["pool1.weight"]
    xo, eo, bo, perm, sc = R.topk_pool(x, ei, batch, w, 0.5)
    gx, ge, _, gb, gperm, gsc = NP.topk_pool(x.to(dev), ei.to(dev), batch.to(dev), w.to(dev), 0.5)
    assert torch.equal(gperm.cpu(), perm)
    assert torch.equal(gb.cpu(), bo)
    assert torch.equal(ge.cpu(), eo)
    assert torch.allclose(gx.cpu(), xo, atol=1e-6, rtol=1e-6)
    assert torch.allclose(gsc.cpu(), sc, atol=1e-6)
    nb = int(bo.max()) + 1
    assert torch.allclose(NP.global_max_mean_pool(gx, gb).cpu(), R.readout(xo, bo, nb), atol=1e-6, rtol=1e-6)


def test_topk_large_graph_and_ties(dev):
    g = torch.Generator().manual_seed(0)
    n1, n2 = 5000, 37                                        # 5000 > 1024: the 16384-capacity sort
    x = torch.randn(n1 + n2, 16, generator=g)
    x[100:110] = x[100]                                      # identical rows => identical scores: index order
    batch = torch.cat([torch.zeros(n1, dtype=torch.long), torch.ones(n2, dtype=torch.long)])
    ei = torch.randint(0, n1 + n2, (2, 20000), generator=g)
    w = torch.randn(1, 16, generator=g)
    xo, eo, bo, perm, sc = R.topk_pool(x, ei, batch, w, 0.5)
    gx, ge, _, gb, gperm, gsc = NP.topk_pool(x.to(dev), ei.to(dev), batch.to(dev), w.to(dev), 0.5)
    assert gperm.numel() == 2500 + 19
    assert torch.equal(gperm.cpu(), perm) and torch.equal(ge.cpu(), eo)
    assert torch.allclose(gx.cpu(), xo, atol=1e-6, rtol=1e-6)


def test_topk_graph_beyond_the_lds_sort_capacity(dev):
    """a graph with more than 16,384 nodes (a hub protein's one-hop subgraph at the C4 / C5 scale) used to
    raise NotImplementedError; it now takes the device-wide sort with the same selection rule, forward and backward."""
    g = torch.Generator().manual_seed(2)
    n1, n2 = 20000, 50
    # scores spaced far above fp32 rounding (the CPU oracle and the kernel round tanh differently in the last bit, and
    # among 20,000 random scores some pairs are that close): column 0 carries a shuffled ramp, w points along it
    x = torch.randn(n1 + n2, 16, generator=g) * 0.1
    x[:, 0] = torch.linspace(-2.5, 2.5, n1 + n2)[torch.randperm(n1 + n2, generator=g)]
    x[500:520] = x[500]                                      # ties: lower index first
    batch = torch.cat([torch.zeros(n1, dtype=torch.long), torch.ones(n2, dtype=torch.long)])
    ei = torch.randint(0, n1 + n2, (2, 60000), generator=g)
    w = torch.zeros(1, 16)
    w[0, 0] = 1.0
    xo, eo, bo, perm, sc = R.topk_pool(x, ei, batch, w, 0.5)
    xd = x.to(dev).requires_grad_(True)
    wd = w.to(dev).requires_grad_(True)
    gx, ge, _, gb, gperm, gsc = NP.topk_pool(xd, ei.to(dev), batch.to(dev), wd, 0.5)
    assert gperm.numel() == 10000 + 25
    assert torch.equal(gperm.cpu(), perm) and torch.equal(ge.cpu(), eo) and torch.equal(gb.cpu(), bo)
    assert torch.allclose(gx.detach().cpu(), xo, atol=1e-6, rtol=1e-6)
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    R.topk_pool(xr, ei, batch, wr, 0.5)[0].pow(2).sum().backward()
    gx.pow(2).sum().backward()
    assert torch.allclose(xd.grad.cpu(), xr.grad, atol=1e-4, rtol=1e-4)
    x6, w6 = x.double().requires_grad_(True), w.double().requires_grad_(True)
    R.topk_pool(x6, ei, batch, w6, 0.5)[0].pow(2).sum().backward()
    if torch.equal(R.topk_pool(x.double(), ei, batch, w.double(), 0.5)[4], perm):     # the fp64 run keeps the same nodes
        assert float((wd.grad.double().cpu() - w6.grad).abs().max() / w6.grad.abs().max()) <= 1e-5
    assert torch.allclose(wd.grad.cpu(), wr.grad, atol=1e-4 * float(wr.grad.abs().max()), rtol=1e-4)


@pytest.mark.parametrize("sizes", [[20000, 50], [7, 200_000, 1, 30_000, 16385, 16384, 3], [17000] * 5])
@pytest.mark.parametrize("ratio", [0.5, 0.25])          # (binary fractions: the oracle's float64 ceil and the kernel's float32 ceil agree)
def test_radix_selection_for_graphs_of_any_size(dev, sizes, ratio):
    """graphs above the LDS sort's 16,384 nodes are selected ON THE DEVICE (npi_topk_select_sorted: two
    stable radix sorts, by score and by graph id) -- no torch.argsort, no host read when the sizes are known.  The kept
    nodes, their order (score descending, lower index first among equals), the pooled batch vector and the filtered edge
    list equal the oracle's; the no-sync path (GraphBatch.sizes + padded_edges) gives the same and makes no device read."""
    g = torch.Generator().manual_seed(sum(sizes))
    n = sum(sizes)
    batch = torch.repeat_interleave(torch.arange(len(sizes)), torch.tensor(sizes))
    x = torch.randn(n, 8, generator=g) * 0.1
    x[:, 0] = torch.linspace(-2.5, 2.5, n)[torch.randperm(n, generator=g)]        # scores far apart (see the test above)
    for lo in range(100, n - 40, max(n // 7, 1)):
        x[lo:lo + 20] = x[lo]                                                     # runs of exact ties inside every large graph
    w = torch.zeros(1, 8)
    w[0, 0] = 1.0
    ei = torch.randint(0, n, (2, 3 * n), generator=g)
    xo, eo, bo, perm, sc = R.topk_pool(x, ei, batch, w, ratio)
    got = NP.topk_pool(x.to(dev), ei.to(dev), batch.to(dev), w.to(dev), ratio)                       # sizes unknown: one read
    assert torch.equal(got[4].cpu(), perm) and torch.equal(got[3].cpu(), bo) and torch.equal(got[1].cpu(), eo)
    assert torch.allclose(got[0].cpu(), xo, atol=1e-6, rtol=1e-6)
    xd, eid, wd = x.to(dev), ei.to(dev), w.to(dev)
    gb = NG.GraphBatch(xd, eid, batch.to(dev), sizes=torch.tensor(sizes))
    _pool_tuple(gb, wd, ratio, padded_edges=True)                                                      # warm-up (lazy inits)
    gb = NG.GraphBatch(xd, eid, batch.to(dev), sizes=torch.tensor(sizes))                            # (segment starts not kept)
    torch.cuda.synchronize()
    torch.cuda.set_sync_debug_mode("error")
    try:
        ns = _pool_tuple(gb, wd, ratio, padded_edges=True)
    finally:
        torch.cuda.set_sync_debug_mode("default")
    e = eo.size(1)
    assert torch.equal(ns[4].cpu(), perm) and torch.equal(ns[3].cpu(), bo)
    assert ns[1].size(1) == ei.size(1) and torch.equal(ns[1][:, :e].cpu(), eo) and bool((ns[1][:, e:] == -1).all())
    # a NaN score sorts first, as in torch.sort(descending=True)
    x2 = x.clone()
    x2[sizes[0] // 2] = float("nan")
    ref2 = R.topk_pool(x2, ei, batch, w, ratio)
    got2 = NP.topk_pool(x2.to(dev), eid, batch.to(dev), wd, ratio)
    assert torch.equal(got2[4].cpu(), ref2[3])


def _run_net1(dev, fx, n_graphs):
    model = Net1(fx["x"].size(1)).to(dev)
    model.load_state_dict({k: v.to(dev) for k, v in fx["state_dict"].items()})      # reference checkpoint, unchanged
    model.eval()
    return model(fx["x"].to(dev), fx["edge_index"].to(dev), fx["batch"].to(dev)).detach().cpu()


def test_whole_net1_on_gpu_reproduces_reference_log_rpi369(dev):
    """result/1228_1/log_0.txt: Accuracy 0.62838 ... = TP 42 FN 32 TN 51 FP 23, with every layer of the
    reference model (convs, pooling, readout) on the MI355X."""
    fx = load("rpi369_fold0.pt")
    logp = _run_net1(dev, fx, fx["y"].numel())
    cm = kat.confusion(logp, fx["y"])
    assert cm == (42, 32, 51, 23)
    assert ["%.5f" % v for v in R.metrics_from_confusion(*cm)] == fx["logged_metrics"]
    assert torch.allclose(logp, fx["logp"], atol=1e-4, rtol=1e-4)


def test_whole_net1_on_gpu_reproduces_case_study_probabilities(dev):
    fx = load("npinter2_katp.pt")
    logp = _run_net1(dev, fx, fx["p_positive_logged"].numel())
    err = (logp[:, 1].double().exp() - fx["p_positive_logged"]).abs().max()
    assert float(err) <= 1e-5


def _batch_case(seed, sizes, F):
    g = torch.Generator().manual_seed(seed)
    batch = torch.cat([torch.full((n,), b, dtype=torch.long) for b, n in enumerate(sizes)])
    N = batch.numel()
    x = torch.randn(N, F, generator=g)
    w = torch.randn(1, F, generator=g)
    starts = torch.cumsum(torch.tensor([0] + list(sizes[:-1])), 0)
    src, dst = [], []
    for b, n in enumerate(sizes):
        if n == 0:
            continue
        e = torch.randint(0, n, (2, 3 * n), generator=g) + starts[b]
        src.append(e[0]); dst.append(e[1])
    ei = torch.stack([torch.cat(src), torch.cat(dst)])
    return x, ei, batch, w


@pytest.mark.parametrize("sizes,F", [((7, 1, 30, 2, 65), 128), ((300, 5), 64), ((40,), 178)])
def test_topk_pool_backward_matches_oracle_autograd(dev, sizes, F):
    x, ei, batch, w = _batch_case(11, sizes, F)
    g = torch.Generator().manual_seed(5)
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    xo, eo, bo, perm, sc = R.topk_pool(xr, ei, batch, wr, 0.5)
    go, gs = torch.randn(xo.shape, generator=g), torch.randn(sc.shape, generator=g)
    (xo * go).sum().add((sc * gs).sum()).backward()
    xg, wg = x.to(dev).requires_grad_(True), w.to(dev).requires_grad_(True)
    gx, ge, _, gb, gperm, gsc = NP.topk_pool(xg, ei.to(dev), batch.to(dev), wg, 0.5)
    assert torch.equal(gperm.cpu(), perm)
    ((gx * go.to(dev)).sum() + (gsc * gs.to(dev)).sum()).backward()
    assert torch.allclose(xg.grad.cpu(), xr.grad, atol=1e-5, rtol=1e-4)
    assert torch.allclose(wg.grad.cpu(), wr.grad, atol=1e-4, rtol=1e-4)
    dropped = torch.ones(x.size(0), dtype=torch.bool)
    dropped[perm] = False
    assert float(xg.grad.cpu()[dropped].abs().max()) == 0.0          # the selection is not differentiable


def test_readout_backward_matches_oracle_autograd(dev):
    x, ei, batch, w = _batch_case(3, (9, 1, 33, 120), 128)
    nb = 4
    xr = x.clone().requires_grad_(True)
    go = torch.randn(nb, 256, generator=torch.Generator().manual_seed(1))
    (R.readout(xr, batch, nb) * go).sum().backward()
    xg = x.to(dev).requires_grad_(True)
    (NP.global_max_mean_pool(xg, batch.to(dev), nb) * go.to(dev)).sum().backward()
    assert torch.allclose(xg.grad.cpu(), xr.grad, atol=1e-6, rtol=1e-5)
    # the two halves as the reference calls them (gmp, gap): gradients add up
    xg2 = x.to(dev).requires_grad_(True)
    r = torch.cat([NP.global_max_pool(xg2, batch.to(dev), nb), NP.global_mean_pool(xg2, batch.to(dev), nb)], dim=1)
    (r * go.to(dev)).sum().backward()
    assert torch.allclose(xg2.grad.cpu(), xr.grad, atol=1e-6, rtol=1e-5)


@pytest.mark.parametrize("sizes,F,view", [
    ((9, 1, 33, 120), 128, False),          # 16-byte column groups
    ((2500, 0, 3, 1100), 128, False),       # graphs longer than the row lanes, an empty graph in the middle
    ((40, 17), 178, False),                 # F = 178: 8-byte groups
    ((40, 17, 5), 7, False),                # odd width: scalar columns
    ((64, 31), 128, True),                  # a column view of a wider matrix: rows not 16-byte aligned
    ((5, 6), 1300, False),                  # more column groups than one workgroup holds
])
def test_readout_forward_backward_shapes(dev, sizes, F, view):
    g = torch.Generator().manual_seed(17)
    batch = torch.cat([torch.full((n,), b, dtype=torch.long) for b, n in enumerate(sizes)])
    nb = len(sizes)
    N = batch.numel()
    wide = torch.randn(N, F + 1, generator=g)
    x = wide[:, 1:].contiguous()
    go = torch.randn(nb, 2 * F, generator=g)
    xr = x.clone().requires_grad_(True)
    ref = R.readout(xr, batch, nb)
    (ref * go).sum().backward()
    if view:
        wd = wide.clone()
        wd[:, 1:] = x
        xg = wd.to(dev)[:, 1:].requires_grad_(True)
    else:
        xg = x.to(dev).requires_grad_(True)
    out = NP.global_max_mean_pool(xg, batch.to(dev), nb)
    full = torch.tensor([n > 0 for n in sizes])
    assert torch.allclose(out.detach().cpu()[full], ref.detach()[full], atol=2e-6, rtol=1e-5)
    assert float(out.detach().cpu()[~full].abs().sum()) == 0.0      # an empty graph reads out zeros, as torch_scatter fills
    (out * go.to(dev)).sum().backward()
    assert torch.allclose(xg.grad.cpu(), xr.grad, atol=1e-6, rtol=1e-5)


def test_pooled_batch_carries_its_segment_starts(dev):
    """The pooled GraphBatch carries the kept-row offsets as its segment starts; they must equal a fresh search."""
    x, ei, batch, w = _batch_case(4, (7, 0, 30, 2, 65, 1), 128)
    out, _, _ = NP.topk_pool_batch(NG.GraphBatch(x.to(dev), ei.to(dev), batch.to(dev), 6), w.to(dev), 0.5)
    assert out.graph_ptr is not None and out.segment_ptr() is out.graph_ptr and out.num_graphs == 6
    fresh = NP.graph_ptr(out.batch, 6)
    assert torch.equal(out.graph_ptr.cpu(), fresh.cpu())
    assert torch.equal(NP.graph_ptr(out.batch, 9).cpu()[:7], fresh.cpu())     # another graph count: searched again
    # the readout of the GraphBatch = the readout of its tensors
    assert torch.equal(NP.global_max_mean_pool(out), NP.global_max_mean_pool(out.x, out.batch, 6))


def test_readout_max_gradient_goes_to_the_first_of_tied_rows(dev):
    """torch_scatter's scatter_max backward routes the gradient to ONE arg-max row; ties resolve to the lowest row."""
    x = torch.randn(700, 128, generator=torch.Generator().manual_seed(2))
    x[650] = x.max(dim=0).values + 1.0                            # the column maxima, three times
    x[40] = x[650]
    x[333] = x[650]
    batch = torch.zeros(700, dtype=torch.long)
    xg = x.to(dev).requires_grad_(True)
    NP.global_max_pool(xg, batch.to(dev), 1).sum().backward()
    expect = torch.zeros_like(x)
    expect[40] = 1.0
    assert torch.equal(xg.grad.cpu(), expect)


def test_net1_training_step_gradients_match_oracle(dev):
    """One step of the reference's train loop (src/train_with_twoDataset.PY:49-55: nll_loss on the
    log-softmax output, backward) on the RPI369 fold-0 batch with the reference checkpoint: every
    parameter gradient of Net_1 -- convs, TopKPooling weights, MLP -- from the MI355X path against
    autograd through the oracle."""
    fx = load("rpi369_fold0.pt")
    y = fx["y"].long()
    sd = {k: v.clone().requires_grad_(True) for k, v in fx["state_dict"].items()}
    logp_ref = R.net1_forward(sd, fx["x"], fx["edge_index"], fx["batch"], y.numel())
    loss_ref = F.nll_loss(logp_ref, y)
    loss_ref.backward()
    model = Net1(fx["x"].size(1)).to(dev)
    model.load_state_dict({k: v.to(dev) for k, v in fx["state_dict"].items()})
    model.train()
    model.dropout_p = 0.0                                     # the oracle restates the eval-mode wiring (no dropout)
    logp = model(fx["x"].to(dev), fx["edge_index"].to(dev), fx["batch"].to(dev))
    loss = F.nll_loss(logp, y.to(dev))
    loss.backward()
    assert abs(float(loss.detach()) - float(loss_ref.detach())) < 1e-5
    for name, prm in model.named_parameters():
        ref = sd[name].grad
        assert prm.grad is not None, name
        scale = float(ref.abs().max()) + 1e-12
        err = float((prm.grad.cpu() - ref).abs().max()) / scale
        assert err < 2e-4, (name, err, scale)
    # and an optimiser step moves the pooling weights too
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    before = model.pool1.weight.detach().clone()
    opt.step()
    assert not torch.equal(before, model.pool1.weight.detach())


def test_pooling_without_a_device_read_when_graph_sizes_are_known(dev):
    """with the per-graph node counts known on the host (GraphBatch.sizes, what net1.KeyLoader supplies)
    TopKPooling reads nothing back: the kept counts are ceil(ratio n_g), the surviving edges stay in an array of the
    input's length padded with (-1, -1) columns that the CSR build and the next filter_adj drop.  Same numbers as the
    path that reads its sizes, forward and backward, and torch's sync detector stays quiet over a whole Net_1 step."""
    from npi_gnn_amd import net1
    fx = load("rpi369_fold0.pt")
    x, ei, batch, y = (fx[k].to(dev) for k in ("x", "edge_index", "batch", "y"))
    B = y.numel()
    sizes = torch.bincount(fx["batch"], minlength=B)
    w = torch.randn(1, x.size(1), generator=torch.Generator().manual_seed(3)).to(dev)
    ref = NP.topk_pool(x, ei, batch, w, 0.5, num_graphs=B)
    plain = _pool_tuple(NG.GraphBatch(x, ei, batch, sizes=sizes), w, 0.5)   # sizes known but padding not asked for: PyG's contract
    assert plain[1].size(1) == ref[1].size(1) and torch.equal(plain[1], ref[1])
    pooled, perm, _ = NP.topk_pool_batch(NG.GraphBatch(x, ei, batch, sizes=sizes), w, 0.5, padded_edges=True)
    got = (pooled.x, pooled.edge_index, None, pooled.batch, perm)
    e = ref[1].size(1)
    assert got[1].size(1) == ei.size(1) and torch.equal(got[1][:, :e], ref[1]) and bool((got[1][:, e:] == -1).all())
    assert torch.equal(got[0], ref[0]) and torch.equal(got[3], ref[3]) and torch.equal(got[4], ref[4])
    assert torch.equal(pooled.sizes, torch.ceil(0.5 * sizes.float()).long())
    # a conv over the padded edge list = a conv over the compact one
    conv = npi.SAGEConv(x.size(1), 32).to(dev)
    assert torch.equal(conv(got[0], got[1]), conv(ref[0], ref[1]))
    # whole Net_1 training step: identical loss and gradients, and no synchronising call at all
    torch.manual_seed(0)
    model = net1.Net_1(x.size(1)).to(dev)
    model.eval()                                              # (dropout off: the two runs must be comparable)

    def run(known):
        model.zero_grad(set_to_none=True)
        data = net1.Batch(NG.GraphBatch(x, ei, batch), y, sizes=known)
        loss = torch.nn.functional.nll_loss(model(data), y)
        loss.backward()
        return loss.detach().clone(), [p.grad.clone() for p in model.parameters()]
    l0, g0 = run(None)
    run(sizes)                                                # warm-up: lazy initialisations may synchronise once
    torch.cuda.synchronize()
    torch.cuda.set_sync_debug_mode("error")
    try:
        l1, g1 = run(sizes)
    finally:
        torch.cuda.set_sync_debug_mode("default")
    assert torch.equal(l0, l1) and all(torch.equal(a, b) for a, b in zip(g0, g1))


def test_padded_edge_lists_edge_cases(dev):
    """The no-read pooling path on degenerate inputs: single-node graphs, a batch whose every edge is dropped (the next conv
    then sees nothing but padding and aggregates each node with itself), ratio 1.0, and an empty key list for the sizes query."""
    from npi_gnn_amd.subgraph import InteractionGraph
    g = torch.Generator().manual_seed(9)
    sizes = torch.tensor([1, 2, 5, 1, 7])
    n = int(sizes.sum())
    batch = torch.repeat_interleave(torch.arange(5), sizes)
    x = torch.randn(n, 12, generator=g)
    # edges only between the two nodes of graph 1: with ratio 0.5 one of them is dropped -> no edge survives
    ei = torch.tensor([[1, 2], [2, 1]])
    w = torch.randn(1, 12, generator=g)
    ref = R.topk_pool(x, ei, batch, w, 0.5)
    pooled, perm, _ = NP.topk_pool_batch(NG.GraphBatch(x.to(dev), ei.to(dev), batch.to(dev), sizes=sizes), w.to(dev), 0.5,
                                         padded_edges=True)
    got = (pooled.x, pooled.edge_index, None, pooled.batch, perm)
    assert torch.equal(got[4].cpu(), ref[3]) and got[1].shape == (2, 2) and bool((got[1] == -1).all()) and ref[1].numel() == 0
    assert torch.equal(pooled.sizes, torch.tensor([1, 1, 3, 1, 4]))
    conv = npi.SAGEConv(12, 8).to(dev)
    out = conv(got[0], got[1])                                   # only padding: every node aggregates itself
    assert torch.allclose(out, got[0] @ conv.weight + conv.bias, atol=1e-5)
    # second pooling layer on the padded list, ratio 1.0 keeps everything
    got2, _, _ = NP.topk_pool_batch(pooled, torch.randn(1, 12, generator=g).to(dev), 1.0, padded_edges=True)
    assert got2.x.size(0) == got[0].size(0) and bool((got2.edge_index == -1).all())
    # sizes of an empty key list; size hints for one key
    ig = InteractionGraph(torch.tensor([[0, 3], [1, 3], [1, 4]]).to(dev), torch.tensor([True, True, True]).to(dev),
                          torch.randn(5, 6, generator=g).to(dev))
    nodes, pairs = ig.sizes(torch.zeros((0, 2), dtype=torch.long, device=dev))
    assert nodes.numel() == 0 and pairs.numel() == 0
    nodes, pairs = ig.sizes(torch.tensor([[1, 3]], device=dev))
    xb, eb, bb = ig.batch(torch.tensor([[1, 3]], device=dev), n_nodes=int(nodes[0]), n_pairs=int(pairs[0]))
    assert xb.size(0) == int(nodes[0]) == 4 and eb.size(1) == 2 * int(pairs[0]) == 6        # rna 1, protein 3, rna 0, protein 4


def test_pooled_csr_derived_from_the_parent_equals_a_fresh_build(dev):
    """graph.filtered_side (npi_csr_filter): the by-target CSR of the pooled graph, derived from the parent's without a sort,
    equals build_side on the filtered (padded) edge list entry for entry -- rowptr, col, eid, rowidx, item_row -- over two
    pooling layers; and the conv that follows finds it in the GraphBatch instead of sorting."""
    from npi_gnn_amd import graph as NG
    fx = torch.load(os.path.join(G, "rpi369_extract.pt"), map_location="cpu", weights_only=False)
    from npi_gnn_amd.subgraph import InteractionGraph
    from npi_gnn_amd import net1
    ig = InteractionGraph(fx["pairs"].long().to(dev), fx["usable"].to(dev), fx["feat"].to(dev))
    keys = fx["keys"].long().to(dev)
    loader = net1.KeyLoader(ig, keys, torch.zeros(keys.size(0), dtype=torch.long, device=dev), 64)
    gb = next(iter(loader))
    torch.manual_seed(0)
    w = torch.randn(1, gb.x.size(1), device=dev)
    assert gb.peek_graph() is None
    gb.graph()                                                   # what the conv in front of the pool does
    assert gb.peek_graph() is not None and gb.symmetric and gb.sizes is not None
    for layer in range(2):
        out, perm, _ = NP.topk_pool_batch(gb.with_x(gb.x.clone().requires_grad_(layer == 1)), w, 0.5, padded_edges=True)
        xo, eo = out.x, out.edge_index
        assert out._recipe is not None and out.peek_graph() is None          # derived on demand
        g = NG.as_graph(out, xo.size(0))
        assert g is out.peek_graph() and out._recipe is None and g.num_nodes == xo.size(0) and g.symmetric
        ref = NG.build_side(eo[1].contiguous(), eo[0].contiguous(), xo.size(0), xo.size(0))
        nnz = int(ref.rowptr[-1])
        assert g.by_dst.nnz_max == ref.nnz_max and g.by_dst.n_items == ref.n_items
        assert torch.equal(g.by_dst.rowptr, ref.rowptr)
        for a, r in ((g.by_dst.col, ref.col), (g.by_dst.eid, ref.eid), (g.by_dst.rowidx, ref.rowidx)):
            assert torch.equal(a[:nnz], r[:nnz])
        assert torch.equal(g.by_dst.item_row, ref.item_row)
        assert out.graph() is g and out.with_x(xo.detach()).graph() is g      # the next conv (and the pool behind it) sort nothing
        gb = out.with_x(xo.detach())


@pytest.mark.parametrize("E", [3000, 2048 * 2048, 2048 * 2048 + 5000])
def test_filter_adj_keeps_the_edge_order_on_both_tile_paths(dev, E):
    """npi_filter_adj(_ex): the surviving edges, renumbered, in their original order, and the position map in the workspace --
    with the tile offsets summed inside the write kernel (up to 2,048 tiles of 2,048 edges) and with the scan launch in
    between (above): against a torch mask / cumsum"""
    import ctypes  # noqa: F401
    from npi_gnn_amd._lib import check, load, ptr, stream_ptr
    lib = load()
    g = torch.Generator().manual_seed(E)
    N = 50_000
    src = torch.randint(0, N, (E,), generator=g).to(dev)
    dst = torch.randint(0, N, (E,), generator=g).to(dev)
    keep = (torch.rand(N, generator=g) < 0.6).to(dev)
    remap = torch.where(keep, torch.cumsum(keep.int(), 0).int() - 1, torch.full((N,), -1, dtype=torch.int32, device=dev)).int()
    out = torch.empty((2, E), dtype=torch.int64, device=dev)
    count = torch.empty(1, dtype=torch.int32, device=dev)
    ws = torch.empty(int(lib.npi_filter_adj_workspace_elems(E)), dtype=torch.int32, device=dev)
    check(lib.npi_filter_adj(ptr(src), ptr(dst), E, ptr(remap), ptr(out[0]), ptr(out[1]), ptr(count), ptr(ws), 1,
                                stream_ptr(dev)), "npi_filter_adj")
    m = keep[src] & keep[dst]
    n = int(m.sum())
    assert int(count.item()) == n
    assert torch.equal(out[0, :n], remap[src[m]].long()) and torch.equal(out[1, :n], remap[dst[m]].long())
    assert bool((out[:, n:] == -1).all())                                   # pad_tail
    off = int(lib.npi_filter_adj_newpos_offset(E))
    newpos = ws[off:off + E]
    ref = torch.where(m, torch.cumsum(m.int(), 0).int() - 1, torch.full((E,), -1, dtype=torch.int32, device=dev)).int()
    assert torch.equal(newpos, ref)


def test_in_place_edit_of_a_cached_edge_list_is_never_answered_from_the_stale_csr(dev):
    """the CSR a GraphBatch keeps for its edge list (and the recipe TopKPooling leaves in the pooled batch)
    is keyed by the tensor's ``_version``: an in-place write to the list -- same shape, same storage -- makes the next conv
    sort again instead of aggregating over the old adjacency.  Plain tensors carry no state at all."""
    g = torch.Generator().manual_seed(4)
    N, E, F = 300, 2000, 16
    ei = torch.randint(0, N, (2, E), generator=g).to(dev)
    x = torch.randn(N, F, generator=g).to(dev)
    conv = npi.SAGEConv(F, 8).to(dev)
    gb = NG.GraphBatch(x, ei)
    g0 = gb.graph()
    out1 = conv(gb).x
    assert gb.peek_graph() is g0 and gb.graph() is g0          # reused while the tensor is untouched
    assert torch.equal(out1, conv(x, ei)) and not hasattr(ei, "_npi_graph")      # the tensor form: same numbers, no state
    ei[0, :500] = torch.randint(0, N, (500,), generator=g).to(dev)       # in-place: same shape, same storage
    assert gb.peek_graph() is None
    out2 = conv(gb).x
    assert gb.graph() is not g0
    want = conv(x, ei.clone())
    assert torch.equal(out2, want) and not torch.equal(out1, out2)
    # the recipe a pooling layer leaves on the list it returns is dropped the same way
    fx = load("rpi369_fold0.pt")
    x, e, batch, y = (fx[k].to(dev) for k in ("x", "edge_index", "batch", "y"))
    gb = NG.GraphBatch(x, e, batch, sizes=torch.bincount(fx["batch"], minlength=y.numel()), symmetric=True)
    gb.graph()                                                 # the conv in front of the pool
    w = torch.randn(1, x.size(1), generator=g).to(dev)
    out, _, _ = NP.topk_pool_batch(gb, w, 0.5, padded_edges=True)
    xo, eo = out.x, out.edge_index
    assert out._recipe is not None
    keep = eo[0] >= 0
    eo[:, keep] = eo[:, keep].flip(0)                          # in-place edit (here: every edge reversed)
    got = out.graph()
    assert out._recipe is None
    ref = NG.build_side(eo[1].contiguous(), eo[0].contiguous(), xo.size(0), xo.size(0))
    nnz = int(ref.rowptr[-1])
    assert torch.equal(got.by_dst.rowptr, ref.rowptr) and torch.equal(got.by_dst.col[:nnz], ref.col[:nnz])


def test_stale_batch_totals_are_an_error_not_an_out_of_bounds_write(dev):
    """InteractionGraph.batch(n_nodes=, n_pairs=) sizes its outputs from the caller's totals while the kernels
    write at device-computed offsets.  The fill kernels compare the two THEMSELVES, on every call: totals of OTHER keys write
    nothing and raise a status bit that the next device read reports (at once under graph.set_debug) -- no host sync on the
    good path, no out-of-bounds write on the bad one; host totals beyond int32 keep the overflow check."""
    from npi_gnn_amd.subgraph import InteractionGraph
    g = torch.Generator().manual_seed(2)
    pairs = torch.tensor([[0, 3], [1, 3], [1, 4], [2, 4], [0, 4]])
    ig = InteractionGraph(pairs.to(dev), torch.ones(5, dtype=torch.bool).to(dev), torch.randn(5, 6, generator=g).to(dev))
    keys = pairs[:3].to(dev)
    nodes, npairs = ig.sizes(keys)
    tot = dict(n_nodes=int(nodes.sum()), n_pairs=int(npairs.sum()))
    NG.check_pending()
    ok = ig.batch(keys, **tot)
    NG.check_pending()                                                  # the right totals: clean
    assert ok.x.size(0) == int(nodes.sum())
    ref = ig.batch(keys)
    assert torch.equal(ok.x, ref.x) and torch.equal(ok.edge_index, ref.edge_index)
    for wrong in (keys[:2], keys[:1]):                                  # a LATER call is caught like the first one
        with torch.cuda.device(dev):
            guard = torch.full((4096,), 7.0, device=dev)                # a neighbour in the allocator's pool
        bad = ig.batch(wrong, **tot)                                    # totals of three keys: nothing may be written
        with pytest.raises(ValueError):
            NG.check_pending()
        assert bool((guard == 7.0).all()) and bad.x.size(0) == int(nodes.sum())
        ig.batch(keys, **tot)
        NG.check_pending()
    NG.set_debug(True)
    try:
        with pytest.raises(ValueError):
            ig.batch(keys[:1], **tot)
    finally:
        NG.set_debug(False)
    with pytest.raises(OverflowError):
        ig.batch(keys, n_nodes=2 ** 31, n_pairs=5)


def test_net1_takes_a_pyg_style_batch_as_well_as_a_graph_batch(dev):
    """Net_1.forward(data): ``data`` may be anything with ``x / edge_index / batch / num_graphs`` (what a PyG DataLoader
    yields) -- it is wrapped into a GraphBatch WITHOUT the optional knowledge -- or a GraphBatch that carries it (host sizes,
    symmetric edge list): the same log-probabilities up to the summation order, and gradients reach every parameter."""
    import types
    from npi_gnn_amd import net1
    fx = load("rpi369_fold0.pt")
    x, ei, batch, y = (fx[k].to(dev) for k in ("x", "edge_index", "batch", "y"))
    B = y.numel()
    torch.manual_seed(0)
    model = net1.Net_1(x.size(1)).to(dev)
    model.eval()
    plain = types.SimpleNamespace(x=x, edge_index=ei, batch=batch, num_graphs=B, y=y)
    rich = net1.Batch(NG.GraphBatch(x, ei, batch, symmetric=True), y, sizes=torch.bincount(fx["batch"], minlength=B))
    a = model(plain)
    b = model(rich)
    assert a.shape == (B, 2) and torch.allclose(a, b, atol=1e-5, rtol=1e-5)
    assert torch.equal(a.argmax(1), b.argmax(1))
    model.train()
    torch.nn.functional.nll_loss(model(plain), y).backward()
    assert all(p.grad is not None and bool(torch.isfinite(p.grad).all()) for p in model.parameters())
    assert not hasattr(ei, "_npi_graph") and not hasattr(batch, "_npi_sizes")        # nothing was left on the caller's tensors
