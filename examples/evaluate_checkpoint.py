#!/usr/bin/env python3
"""The reference's evaluation flow (src/test.py: load a saved model, run the test set, print the confusion matrix and the
five metrics) on one MI355X, with the reference's OWN checkpoint: ``Net_1.load_state_dict`` takes the state dict that
``torch.save(model.state_dict(), 'result/<proj>/model_<k>_fold/<epoch>')`` wrote, unchanged -- parameter names and layouts
are PyG 1.4.2's.  NPInter2 (project 1223_1), fold 0: the authors logged TP 1994 / FN 89 / TN 1901 / FP 182
(result/1223_1/log_0.txt); this prints the same four numbers.

Data: tests/golden/npinter2_folds.pt (the fold's keys, features and the reference checkpoint; made by
tests/golden/make_npinter2_folds.py from the reference's data files).

usage: python examples/evaluate_checkpoint.py [--fold 0] [--batch 200] [--checkpoint path/to/state_dict.pt]
"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from npi_gnn_amd import metrics as NM, net1  # noqa: E402
from train_npinter2 import load_fold  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--fold", type=int, default=0)
    ap.add_argument("--batch", type=int, default=200)
    ap.add_argument("--checkpoint", default=None, help="a state dict saved by the reference (default: the fixture's fold-0 one)")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    ig, _, _, test_keys, test_y, F_in, fx = load_fold(dev, a.fold)
    sd = torch.load(a.checkpoint, map_location="cpu") if a.checkpoint else fx[f"fold{a.fold}"].get("state_dict")
    if sd is None:
        raise SystemExit(f"the fixture holds no checkpoint for fold {a.fold}; pass --checkpoint")
    model = net1.Net_1(F_in).to(dev)
    model.load_state_dict(sd)                                   # src/test.py:41
    model.eval()
    counts = torch.zeros(4, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    t0 = time.time()
    with torch.no_grad():
        for data in net1.KeyLoader(ig, test_keys, test_y, a.batch):
            NM.confusion_update(model(data), data.y, counts)
    tp, fn, tn, fp = counts.tolist()
    dt = time.time() - t0
    print(f"TP: {tp}, FN: {fn}, TN: {tn}, FP: {fp}")
    m = NM.Accuracy_Precision_Sensitivity_Specificity_MCC(model, net1.KeyLoader(ig, test_keys, test_y, a.batch), dev)
    print("Accuracy: {:.5f}, Precision: {:.5f}, Sensitivity: {:.5f}, Specificity: {:.5f}, MCC: {:.5f}".format(*m))
    logged = fx[f"fold{a.fold}"].get("confusion_TP_FN_TN_FP")
    if logged is not None and a.checkpoint is None:
        print(f"reference log: TP/FN/TN/FP = {logged} -> {'identical' if logged == [tp, fn, tn, fp] else 'DIFFERENT'}")
    print(f"{test_keys.size(0)} enclosing subgraphs extracted and classified in {dt:.2f} s")


if __name__ == "__main__":
    main()
