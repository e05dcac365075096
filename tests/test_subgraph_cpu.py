"""oracle/ref_subgraph.py (enclosing-subgraph extraction + collate) against the
committed golden vectors of the reference's RPI369 project, and -- where /root/reference exists --
against the KAT-pinned oracle/kat.py on the live data."""
import os

import pytest
import torch

from oracle import kat, ref_subgraph as RS

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def canon(ei):
    k = ei[0].long() * (int(ei.max()) + 1 if ei.numel() else 1) + ei[1].long()
    return ei[:, torch.argsort(k)].long()


def test_oracle_extraction_matches_golden_vectors():
    fx = torch.load(os.path.join(G, "rpi369_extract.pt"), map_location="cpu", weights_only=False)
    pairs, usable, feat = fx["pairs"].long(), fx["usable"], fx["feat"]
    x, ei, b, nid = RS.enclosing_subgraph_batch(pairs, usable, feat, fx["keys"].long())
    assert torch.equal(x, fx["x"]) and torch.equal(b, fx["batch"].long()) and torch.equal(nid, fx["node_id"].long())
    assert torch.equal(canon(ei), fx["edge_index_sorted"].long())
    # the same batch is what the Net_1 fixtures were made from
    net = torch.load(os.path.join(G, "rpi369_fold0.pt"), map_location="cpu", weights_only=False)
    assert torch.equal(x, net["x"]) and torch.equal(canon(ei), canon(net["edge_index"]))
    # corner cases on real data: targets that are usable training pairs, and targets that are no edge
    x2, ei2, b2, nid2 = RS.enclosing_subgraph_batch(pairs, usable, feat, fx["keys2"].long())
    assert x2.size(0) == fx["x2_rows"] and torch.equal(nid2, fx["node_id2"].long()) and torch.equal(b2, fx["batch2"].long())
    assert torch.equal(canon(ei2), fx["edge_index2_sorted"].long())
    lab = torch.ones(x2.size(0))
    starts = torch.cat([torch.zeros(1, dtype=torch.long), torch.cumsum(torch.bincount(b2), 0)[:-1]])
    lab[starts] = 0
    lab[starts + 1] = 0
    assert torch.equal(x2, torch.cat([lab.view(-1, 1), feat[nid2]], 1))


def test_oracle_extraction_small_hand_case():
    # rna 0: proteins 10 (usable), 11 (test key -> unusable), 12;  rna 1: protein 10;  rna 2: protein 12
    pairs = torch.tensor([[0, 10], [0, 11], [0, 12], [1, 10], [2, 12]])
    usable = torch.tensor([True, False, True, True, True])
    feat = torch.arange(13.0).view(-1, 1)
    x, ei, b, nid = RS.enclosing_subgraph_batch(pairs, usable, feat, torch.tensor([[0, 11], [1, 12]]))
    # sample 0: target (0, 11) itself unusable but always present; partners of rna 0: 10, 12; partners of protein 11: none usable
    # sample 1: target (1, 12) is no edge; partner of rna 1: 10; partners of protein 12 in list order: rna 0, rna 2
    assert nid.tolist() == [0, 11, 10, 12, 1, 12, 10, 0, 2]
    assert b.tolist() == [0, 0, 0, 0, 1, 1, 1, 1, 1]
    assert x[:, 0].tolist() == [0, 0, 1, 1, 0, 0, 1, 1, 1]
    assert ei.t().tolist() == [[0, 1], [1, 0], [0, 2], [2, 0], [0, 3], [3, 0],
                               [4, 5], [5, 4], [4, 6], [6, 4], [7, 5], [5, 7], [8, 5], [5, 8]]


@pytest.mark.skipif(not kat.have_reference(), reason="needs /root/reference")
def test_oracle_extraction_matches_kat_on_live_npinter2():
    proj = kat.Project("NPInter2", "1223_1", 0)
    pairs = torch.tensor(proj.pos + proj.neg)
    usable = torch.tensor([k not in proj.cannot for k in proj.pos + proj.neg])
    keys = proj.test_pos[:40] + proj.test_neg[:40]
    x, ei, b = proj.batch(keys)
    ox, oe, ob, _ = RS.enclosing_subgraph_batch(pairs, usable, proj.feat.float(), torch.tensor(keys))
    assert torch.equal(ox, x) and torch.equal(ob, b) and torch.equal(canon(oe), canon(ei))
