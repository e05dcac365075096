#!/usr/bin/env python3
"""End-to-end: the reference's training flow (src/train_with_twoDataset.PY:46-57,130-200) on its RPI369 project,
fold 0, with every device-side piece of this package -- samples built on the GPU from target pairs
(InteractionGraph), Net_1 from SAGEConv / TopKPooling / readout modules, evaluation through the confusion
kernel.  Data: tests/golden/rpi369_extract.pt (the project's pair list, cannotUse mask, node features and
test keys, regenerated from the reference's data files by tests/golden/make_golden_extract.py).

usage: python examples/train_rpi369.py [--epochs 50] [--batch 200]
The loop is npi_gnn_amd.net1.fit -- the reference's, statement by statement: the datasets are shuffled ONCE
(`dataset.shuffle()`, :78-79), the loaders do not shuffle (:142-143), Adam lr 1e-3 / weight decay 1e-3, batch 200,
ExponentialLR(0.95) stepped only in epochs whose loss rose (:158-160), metrics on both loaders every 5th epoch.
"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from npi_gnn_amd import net1  # noqa: E402
from npi_gnn_amd.subgraph import InteractionGraph  # noqa: E402


def load_project(dev):
    fx = torch.load(os.path.join(ROOT, "tests", "golden", "rpi369_extract.pt"), map_location="cpu", weights_only=False)
    pairs, usable = fx["pairs"].long(), fx["usable"]
    n_pos = pairs.size(0) // 2                       # RPI369: 369 positives in xlsx order, then as many negatives
    label = torch.cat([torch.ones(n_pos, dtype=torch.long), torch.zeros(pairs.size(0) - n_pos, dtype=torch.long)])
    ig = InteractionGraph(pairs.to(dev), usable.to(dev), fx["feat"].to(dev))
    train_keys, train_y = pairs[usable].to(dev), label[usable].to(dev)
    test_keys = fx["keys"].long().to(dev)
    code = {(int(a), int(b)): int(l) for (a, b), l in zip(pairs.tolist(), label.tolist())}
    test_y = torch.tensor([code[(int(a), int(b))] for a, b in fx["keys"].tolist()], device=dev)
    return ig, train_keys, train_y, test_keys, test_y, fx["feat"].size(1) + 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--epochs", type=int, default=50)
    ap.add_argument("--batch", type=int, default=200)
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(a.seed)
    ig, train_keys, train_y, test_keys, test_y, F_in = load_project(dev)
    g = torch.Generator().manual_seed(a.seed)
    train_loader = net1.KeyLoader(ig, train_keys, train_y, a.batch).shuffle(g)
    test_loader = net1.KeyLoader(ig, test_keys, test_y, a.batch).shuffle(g)
    print(f"RPI369 fold 0: {train_keys.size(0)} training pairs, {test_keys.size(0)} test pairs, F={F_in}")
    model = net1.Net_1(F_in).to(dev)
    res = net1.fit(model, train_loader, test_loader, dev, num_of_epoch=a.epochs)
    print(f"{a.epochs} epochs in {res['seconds']:.2f} s")


if __name__ == "__main__":
    main()
