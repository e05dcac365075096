"""Generate tests/golden/rpi369_extract.pt (run in the build container, where /root/reference exists):
    python tests/golden/make_golden_extract.py

The interaction graph of reference project 1228_1 (RPI369) fold 0 as tensors -- the ordered pair list
(positives in xlsx order, then the negatives of set_negativeInteractionKey_all), the usable mask
(False for the fold's test keys, src/generate_dataset.py:296-299), node features (node2vec | k-mer)
-- plus the 148 test keys and the batch oracle/kat.py builds for them (node order, x, batch, edges).
oracle/kat.py is the KAT-pinned restatement (it reproduces result/1228_1/log_0.txt exactly), so these
outputs pin oracle/ref_subgraph.py and the device extractor.  Also a second key list that exercises
the corner cases on real data: keys that ARE usable training pairs, and pairs that are no edge at all.
"""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import kat, ref_subgraph as RS  # noqa: E402


def canon(ei):
    k = ei[0] * (int(ei.max()) + 1 if ei.numel() else 1) + ei[1]
    return ei[:, torch.argsort(k)]


def main():
    assert kat.have_reference(), "needs /root/reference"
    proj = kat.Project("RPI369", "1228_1", 0)
    pairs = torch.tensor(proj.pos + proj.neg, dtype=torch.int64)
    usable = torch.tensor([k not in proj.cannot for k in proj.pos + proj.neg])
    feat = proj.feat.to(torch.float32)
    keys = proj.test_pos + proj.test_neg
    x, ei, b = proj.batch(keys)
    node_ids = []
    for l, p in keys:
        node_ids += proj.sample(l, p)[0]
    # corner cases: usable training pairs as targets, and non-edges
    train = [k for k in proj.pos + proj.neg if k not in proj.cannot]
    g = torch.Generator().manual_seed(0)
    rnas = sorted(proj.rna_adj)
    prots = sorted(proj.prot_adj)
    edge_set = set(proj.pos + proj.neg)
    non = []
    while len(non) < 20:
        k = (rnas[int(torch.randint(0, len(rnas), (1,), generator=g))], prots[int(torch.randint(0, len(prots), (1,), generator=g))])
        if k not in edge_set:
            non.append(k)
    keys2 = train[:20] + non
    x2, ei2, b2 = proj.batch(keys2)
    node_ids2 = []
    for l, p in keys2:
        node_ids2 += proj.sample(l, p)[0]
    # the tensor restatement must agree with the KAT-pinned one before anything is written
    for kk, (xx, ee, bb, nn) in ((keys, (x, ei, b, node_ids)), (keys2, (x2, ei2, b2, node_ids2))):
        ox, oe, ob, on = RS.enclosing_subgraph_batch(pairs, usable, feat, torch.tensor(kk))
        assert torch.equal(ox, xx) and torch.equal(ob, bb) and on.tolist() == nn
        assert torch.equal(canon(oe), canon(ee))
    torch.save({"pairs": pairs.to(torch.int32), "usable": usable, "feat": feat, "num_nodes": proj.num_nodes,
                "keys": torch.tensor(keys, dtype=torch.int32), "x": x, "edge_index_sorted": canon(ei).to(torch.int32), "batch": b.to(torch.int32),
                "node_id": torch.tensor(node_ids, dtype=torch.int32),
                "keys2": torch.tensor(keys2, dtype=torch.int32), "x2_rows": x2.size(0), "edge_index2_sorted": canon(ei2).to(torch.int32),
                "batch2": b2.to(torch.int32), "node_id2": torch.tensor(node_ids2, dtype=torch.int32),
                "source": "RPI369.xlsx + data/set_allInteractionKey/1228_1 + node2vec/k-mer files, fold 0; outputs of oracle/kat.py Project.batch"},
               os.path.join(HERE, "rpi369_extract.pt"))
    print("rpi369_extract.pt", pairs.shape, feat.shape, x.shape, ei.shape, x2.shape, ei2.shape,
          os.path.getsize(os.path.join(HERE, "rpi369_extract.pt")))


if __name__ == "__main__":
    main()
