"""Device-side enclosing-subgraph extraction + collate (npi_gnn_amd.subgraph)
against the oracle: bit-exact node order / features / batch vector / edge list."""
import os

import pytest
import torch

import npi_gnn_amd as npi
from npi_gnn_amd import pool as NP
from npi_gnn_amd.subgraph import InteractionGraph
from oracle import kat, ref_conv as R, ref_subgraph as RS

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _check(dev, pairs, usable, feat, keys):
    ig = InteractionGraph(pairs.to(dev), usable.to(dev), feat.to(dev))
    (x, ei, b), nid = ig.batch(keys.to(dev), return_node_id=True)
    ox, oe, ob, on = RS.enclosing_subgraph_batch(pairs, usable, feat, keys)
    assert torch.equal(nid.cpu().long(), on)
    assert torch.equal(b.cpu(), ob)
    assert torch.equal(ei.cpu(), oe)                       # same canonical edge order as the oracle
    assert torch.equal(x.cpu(), ox)                        # bit-exact feature rows
    return x, ei, b


def test_extraction_matches_reference_vectors_rpi369(dev):
    fx = torch.load(os.path.join(G, "rpi369_extract.pt"), map_location="cpu", weights_only=False)
    pairs, usable, feat = fx["pairs"].long(), fx["usable"], fx["feat"]
    x, ei, b = _check(dev, pairs, usable, feat, fx["keys"].long())
    assert torch.equal(x.cpu(), fx["x"])
    _check(dev, pairs, usable, feat, fx["keys2"].long())
    _check(dev, pairs, usable, feat, fx["keys"][:1].long())
    _check(dev, pairs, usable, feat, fx["keys"][:0].long())


def test_extract_then_net1_reproduces_reference_log(dev):
    """keys -> device extraction -> Net_1 on the MI355X = result/1228_1/log_0.txt (TP 42 FN 32 TN 51 FP 23):
    the reference's test flow with no host-side sample construction at all."""
    fx = torch.load(os.path.join(G, "rpi369_extract.pt"), map_location="cpu", weights_only=False)
    net = torch.load(os.path.join(G, "rpi369_fold0.pt"), map_location="cpu", weights_only=False)
    ig = InteractionGraph(fx["pairs"].long().to(dev), fx["usable"].to(dev), fx["feat"].to(dev))
    x, ei, b = ig.batch(fx["keys"].long().to(dev))
    sd = {k: v.to(dev) for k, v in net["state_dict"].items()}
    h, e, bb, acc = x, ei, b, None
    for k in (1, 2, 3):
        h = torch.relu(npi.sage_conv(h, e, sd[f"conv{k}.weight"], sd[f"conv{k}.bias"]))
        h, e, _, bb, _, _ = NP.topk_pool(h, e, bb, sd[f"pool{k}.weight"], 0.5, num_graphs=net["y"].numel())
        r = NP.global_max_mean_pool(h, bb, net["y"].numel())
        acc = r if acc is None else acc + r
    z = torch.relu(torch.nn.functional.linear(acc, sd["lin1.weight"], sd["lin1.bias"]))
    z = torch.relu(torch.nn.functional.linear(z, sd["lin2.weight"], sd["lin2.bias"]))
    logp = torch.log_softmax(torch.nn.functional.linear(z, sd["lin3.weight"], sd["lin3.bias"]), -1).cpu()
    cm = kat.confusion(logp, net["y"])
    assert cm == (42, 32, 51, 23)
    assert ["%.5f" % v for v in R.metrics_from_confusion(*cm)] == net["logged_metrics"]
    assert torch.allclose(logp, net["logp"], atol=1e-4, rtol=1e-4)


@pytest.mark.parametrize("n_rna,n_prot,P,B", [(300, 40, 3000, 200), (50, 3, 140, 64), (2000, 5, 6000, 33)])
def test_extraction_on_synthetic_graphs_with_hubs(dev, n_rna, n_prot, P, B):
    g = torch.Generator().manual_seed(n_rna + P)
    code = torch.randperm(n_rna * n_prot, generator=g)[:P]                 # unique pairs, shuffled list order
    pairs = torch.stack([code // n_prot, n_rna + code % n_prot], 1)
    usable = torch.rand(P, generator=g) > 0.3
    feat = torch.randn(n_rna + n_prot, 37, generator=g)
    keys = torch.cat([pairs[torch.randint(0, P, (B // 2,), generator=g)],          # existing pairs (usable or not)
                      torch.stack([torch.randint(0, n_rna, (B - B // 2,), generator=g),
                                   n_rna + torch.randint(0, n_prot, (B - B // 2,), generator=g)], 1)])
    _check(dev, pairs, usable, feat, keys)


def test_extraction_rejects_bad_graphs(dev):
    feat = torch.zeros(6, 4, device=dev)
    ok = torch.ones(2, dtype=torch.bool, device=dev)
    with pytest.raises(ValueError):
        InteractionGraph(torch.tensor([[0, 3], [0, 3]], device=dev), ok, feat)          # duplicate pair
    with pytest.raises(ValueError):
        InteractionGraph(torch.tensor([[0, 3], [3, 4]], device=dev), ok, feat)          # 3 is rna and protein


def test_per_key_sizes_replace_the_per_batch_device_read(dev):
    """InteractionGraph.sizes: one read for a whole key list; a batch built with the totals it implies is identical to one
    that reads its sizes back (net1.KeyLoader uses this: no host sync per batch)."""
    from npi_gnn_amd import net1
    fx = torch.load(os.path.join(G, "rpi369_extract.pt"), map_location="cpu", weights_only=False)
    ig = InteractionGraph(fx["pairs"].long().to(dev), fx["usable"].to(dev), fx["feat"].to(dev))
    keys = fx["keys"].long().to(dev)
    nodes, pairs = ig.sizes(keys)
    assert nodes.device.type == "cpu" and nodes.numel() == keys.size(0)
    for lo, hi in ((0, 148), (10, 11), (40, 100)):
        x0, e0, b0 = ig.batch(keys[lo:hi])
        assert int(nodes[lo:hi].sum()) == x0.size(0) and 2 * int(pairs[lo:hi].sum()) == e0.size(1)
        x1, e1, b1 = ig.batch(keys[lo:hi], n_nodes=int(nodes[lo:hi].sum()), n_pairs=int(pairs[lo:hi].sum()))
        assert torch.equal(x0, x1) and torch.equal(e0, e1) and torch.equal(b0, b1)
    y = torch.zeros(keys.size(0), dtype=torch.long, device=dev)
    loader = net1.KeyLoader(ig, keys, y, 64).shuffle(torch.Generator().manual_seed(1))
    seen = 0
    for data in loader:
        xs, es, bs = ig.batch(loader.keys[seen:seen + 64])
        assert torch.equal(data.x, xs) and torch.equal(data.edge_index, es)
        seen += data.num_graphs
    assert seen == 148


def test_feature_rows_are_stored_128_aligned_with_zero_pad_columns(dev):
    """an odd feature width is stored with a row pitch of the next multiple of 128: x is the [n, F] view (bit-exact, as the
    tests above check), the pad columns of its buffer are zero, and Net_1's first layer equals the unpadded one"""
    fx = torch.load(os.path.join(G, "rpi369_extract.pt"), map_location="cpu", weights_only=False)
    ig = InteractionGraph(fx["pairs"].long().to(dev), fx["usable"].to(dev), fx["feat"].to(dev))
    keys = fx["keys"].long().to(dev)
    gb = ig.batch(keys)
    x, ei, b = gb
    assert gb.symmetric and gb.num_graphs == keys.size(0)
    F = x.size(1)
    base = gb.pad_base
    if F % 128 != 0 and 2 * ((F + 127) // 128 * 128) <= 3 * F:
        assert base is not None and base.size(1) == (F + 127) // 128 * 128 and base.data_ptr() == x.data_ptr()
        assert x.stride(0) == base.size(1) and not bool(base[:, F:].any())
    gb0 = ig.batch(keys, pad_features=False)
    x0, e0, b0 = gb0
    assert x0.is_contiguous() and gb0.pad_base is None
    assert torch.equal(x, x0) and torch.equal(ei, e0)
    torch.manual_seed(0)
    conv = npi.SAGEConv(F, 128).to(dev)
    go = torch.randn(x.size(0), 128, device=dev)
    outs = []
    for inp in (gb, gb0):                                  # the padded batch runs its GEMMs on the 128-aligned buffer
        conv.zero_grad()
        out = conv(inp).x
        out.backward(go)
        outs.append((out.detach().clone(), conv.weight.grad.clone(), conv.bias.grad.clone()))
    for p, q in zip(*outs):
        torch.testing.assert_close(p, q, atol=2e-6 * float(q.abs().max()) + 1e-6, rtol=1e-5)
