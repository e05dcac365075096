#!/usr/bin/env python3
"""End-to-end: the reference's training flow (src/train_with_twoDataset.PY:46-57,130-200) on its RPI369 project,
fold 0, with every device-side piece of this package -- samples built on the GPU from target pairs
(InteractionGraph), Net_1 from SAGEConv / TopKPooling / readout modules, evaluation through the confusion
kernel.  Data: tests/golden/rpi369_extract.pt (the project's pair list, cannotUse mask, node features and
test keys, regenerated from the reference's data files by tests/golden/make_golden_extract.py).

usage: python examples/train_rpi369.py [--epochs 50] [--batch 200]
Hyper-parameters are the reference's defaults: Adam lr 1e-3, weight decay 1e-3, batch 200, lr x 0.95 per epoch.
"""
import argparse
import os
import sys
import time

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import npi_gnn_amd as npi  # noqa: E402
from npi_gnn_amd import metrics as NM, pool as NP  # noqa: E402
from npi_gnn_amd.subgraph import InteractionGraph  # noqa: E402


class Net_1(torch.nn.Module):
    """The wiring of reference src/classes.py:45-82, built from this package's modules."""

    def __init__(self, num_node_features, num_of_classes=2):
        super().__init__()
        self.conv1, self.pool1 = npi.SAGEConv(num_node_features, 128), NP.TopKPooling(128, ratio=0.5)
        self.conv2, self.pool2 = npi.SAGEConv(128, 128), NP.TopKPooling(128, ratio=0.5)
        self.conv3, self.pool3 = npi.SAGEConv(128, 128), NP.TopKPooling(128, ratio=0.5)
        self.lin1 = torch.nn.Linear(256, 128)
        self.lin2 = torch.nn.Linear(128, 64)
        self.lin3 = torch.nn.Linear(64, num_of_classes)

    def forward(self, data):
        x, edge_index, batch = data.x, data.edge_index, data.batch
        acc = None
        for conv, pool in ((self.conv1, self.pool1), (self.conv2, self.pool2), (self.conv3, self.pool3)):
            x = F.relu(conv(x, edge_index))
            x, edge_index, _, batch, _, _ = pool(x, edge_index, None, batch)
            r = torch.cat([NP.global_max_pool(x, batch, data.num_graphs), NP.global_mean_pool(x, batch, data.num_graphs)], dim=1)
            acc = r if acc is None else acc + r
        x = F.relu(self.lin1(acc))
        x = F.dropout(x, p=0.5, training=self.training)
        x = F.relu(self.lin2(x))
        return F.log_softmax(self.lin3(x), dim=-1)


class Batch:
    """What a PyG DataLoader yields, as far as Net_1 and the metrics look."""

    def __init__(self, x, edge_index, batch, y):
        self.x, self.edge_index, self.batch, self.y = x, edge_index, batch, y
        self.num_graphs = y.numel()

    def to(self, device):
        return self


class KeyLoader:
    """Batches of target pairs -> batches of enclosing subgraphs, built on the device per step."""

    def __init__(self, ig, keys, y, batch_size, shuffle, seed=0):
        self.ig, self.keys, self.y, self.bs, self.shuffle = ig, keys, y, batch_size, shuffle
        self.dataset = range(keys.size(0))
        self.gen = torch.Generator().manual_seed(seed)

    def __iter__(self):
        n = self.keys.size(0)
        order = torch.randperm(n, generator=self.gen) if self.shuffle else torch.arange(n)
        for i in range(0, n, self.bs):
            idx = order[i:i + self.bs].to(self.keys.device)
            x, ei, b = self.ig.batch(self.keys[idx])
            yield Batch(x, ei, b, self.y[idx])


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
    model = Net_1(F_in).to(dev)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-3)
    sched = torch.optim.lr_scheduler.ExponentialLR(opt, gamma=0.95)
    train_loader = KeyLoader(ig, train_keys, train_y, a.batch, shuffle=True, seed=a.seed)
    test_loader = KeyLoader(ig, test_keys, test_y, a.batch, shuffle=False)
    print(f"RPI369 fold 0: {train_keys.size(0)} training pairs, {test_keys.size(0)} test pairs, F={F_in}")
    t0 = time.perf_counter()
    for epoch in range(a.epochs):
        model.train()
        loss_all = 0.0
        for data in train_loader:
            opt.zero_grad()
            loss = F.nll_loss(model(data), data.y)
            loss.backward()
            loss_all += data.num_graphs * float(loss.detach())
            opt.step()
        sched.step()
        if (epoch + 1) % 5 == 0 or epoch == a.epochs - 1:
            m = NM.Accuracy_Precision_Sensitivity_Specificity_MCC(model, test_loader, dev)
            print("Epoch: {:03d}, loss {:.4f}, testing dataset, Accuracy: {:.5f}, Precision: {:.5f}, Sensitivity: {:.5f}, "
                  "Specificity: {:.5f}, MCC: {:.5f}".format(epoch + 1, loss_all / train_keys.size(0), *m))
    torch.cuda.synchronize()
    print(f"{a.epochs} epochs in {time.perf_counter() - t0:.2f} s")


if __name__ == "__main__":
    main()
