#!/usr/bin/env python3
"""The reference's real workload end to end on one MI355X: project 1223_1 (NPInter2), fold 0 -- 16,658 training and
4,166 test subgraphs, batch 200, 50 epochs (src/train_with_twoDataset.PY; result/1223_1/log_0.txt: 1413.5 s on the
authors' machine, test accuracy 0.93495 at epoch 50, 0.930-0.940 over the five folds).

Everything device-side: samples are extracted from target pairs per batch (InteractionGraph), Net_1 is built from
SAGEConv / TopKPooling / readout of this package, evaluation goes through the confusion kernel.  The loop is
npi_gnn_amd.net1.fit, which follows the reference statement by statement: `dataset.shuffle()` once, a loader that does
not shuffle, Adam(1e-3, weight decay 1e-3), ExponentialLR(0.95) stepped only when the epoch loss rose, metrics on the
training AND the test loader every 5th epoch and at the end.
Data: tests/golden/npinter2_folds.pt (made by tests/golden/make_npinter2_folds.py from the reference's data files).

usage: python examples/train_npinter2.py [--epochs 50] [--batch 200] [--seed 0] [--fold 0]
"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from npi_gnn_amd import net1  # noqa: E402
from npi_gnn_amd.subgraph import InteractionGraph  # noqa: E402


def load_fold(dev, fold=0, path=None):
    """(InteractionGraph, train keys, train y, test keys, test y, num_node_features, fixture) of NPInter2 fold `fold`"""
    fx = torch.load(path or os.path.join(ROOT, "tests", "golden", "npinter2_folds.pt"), map_location="cpu", weights_only=False)
    fb = fx[f"fold{fold}"]
    pairs, label = fx["pairs"].long(), fx["label"].long()
    N = fx["num_nodes"]
    test = torch.cat([fb["test_pos"], fb["test_neg"]]).long()
    test_y = torch.cat([torch.ones(fb["test_pos"].size(0), dtype=torch.long), torch.zeros(fb["test_neg"].size(0), dtype=torch.long)])
    code = pairs[:, 0] * N + pairs[:, 1]
    usable = ~torch.isin(code, test[:, 0] * N + test[:, 1])                # src/generate_dataset.py:296-299
    feat = torch.cat([fb["node2vec"], fx["kmer"]], dim=1)
    ig = InteractionGraph(pairs.to(dev), usable.to(dev), feat.to(dev), num_nodes=N)
    return ig, pairs[usable].to(dev), label[usable].to(dev), test.to(dev), test_y.to(dev), feat.size(1) + 1, fx


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--epochs", type=int, default=50)
    ap.add_argument("--batch", type=int, default=200)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--fold", default="0", choices=["0", "1", "2", "3", "4", "all"],
                    help="fold of the reference's five-fold cross-validation; `all` runs the five one after the other")
    ap.add_argument("--no-train-eval", action="store_true", help="skip the metrics on the training loader")
    ap.add_argument("--capture", action="store_true",
                    help="replay every batch's training step from a HIP graph after the first epoch (net1.GraphedEpoch)")
    ap.add_argument("--json", action="store_true", help="print a one-line JSON summary at the end")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    folds = range(5) if a.fold == "all" else [int(a.fold)]
    summary = []
    for fold in folds:
        torch.manual_seed(a.seed + fold)
        ig, train_keys, train_y, test_keys, test_y, F_in, fx = load_fold(dev, fold)
        g = torch.Generator().manual_seed(a.seed + fold)
        train_loader = net1.KeyLoader(ig, train_keys, train_y, a.batch).shuffle(g)         # train_dataset.shuffle()
        test_loader = net1.KeyLoader(ig, test_keys, test_y, a.batch).shuffle(g)            # test_dataset.shuffle()
        print(f'fold {fold}: number of samples in testing dataset：', len(test_loader.dataset),
              'number of samples in training dataset：', len(train_loader.dataset))
        model = net1.Net_1(F_in, 2).to(dev)
        res = net1.fit(model, train_loader, test_loader, dev, num_of_epoch=a.epochs, eval_train=not a.no_train_eval,
                       capture=a.capture)
        logged = fx[f"fold{fold}"].get("logged_metrics")
        summary.append({"fold": fold, "seconds": res["seconds"], "test_acc": res["test"][0], "test_mcc": res["test"][4],
                        "lr_steps": res["lr_steps"], "reference_test_acc": float(logged[0]) if logged else None})
        del model, ig, train_loader, test_loader
        torch.cuda.empty_cache()
    ref = fx["fold0"]
    lo, hi = min(ref["logged_test_acc_5fold_epoch50"]), max(ref["logged_test_acc_5fold_epoch50"])
    total = sum(s_["seconds"] for s_ in summary)
    mean_acc = sum(s_["test_acc"] for s_ in summary) / len(summary)
    print(f"reference (result/1223_1/log_*.txt): {ref['logged_wall_seconds']:.1f} s for fold 0, test accuracy at epoch 50 "
          f"{lo:.5f}-{hi:.5f} over the five folds (mean {sum(ref['logged_test_acc_5fold_epoch50']) / 5:.5f}); this run: "
          f"{total:.1f} s for {len(summary)} fold(s), test accuracy " + ", ".join(f"{s_['test_acc']:.5f}" for s_ in summary) +
          f" (mean {mean_acc:.5f})")
    if a.json:
        print(json.dumps({"seconds": total, "test_acc": mean_acc, "folds": summary, "epochs": a.epochs, "captured": a.capture,
                          "reference_seconds": ref["logged_wall_seconds"], "reference_acc_range": [lo, hi]}))


if __name__ == "__main__":
    main()
