"""Evaluation metrics (reference src/methods.py:78-126): the host formulas against the
reference's own logged metric lines (tests/golden/kat_expected.json), the confusion kernel against the
reference's per-element rule."""
import json
import os

import pytest
import torch

from npi_gnn_amd import metrics as NM
from oracle import ref_conv as R

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_metric_formulas_reproduce_the_logged_lines():
    exp = json.load(open(os.path.join(G, "kat_expected.json")))
    n = 0
    for proj, runs in exp.items():
        for run, rec in runs.items():
            TP, FN, TN, FP = rec["TP_FN_TN_FP"]
            got = NM.metrics_from_counts(TP, FN, TN, FP)
            assert ["%.5f" % v for v in got] == rec["logged"], (proj, run)
            assert got == R.metrics_from_confusion(TP, FN, TN, FP)
            n += 1
    assert n >= 4
    assert NM.metrics_from_counts(0, 0, 0, 0) == (0, 0, 0, 0, 0)


def _reference_loop(scores, y):
    TP = TN = FP = FN = 0
    pred = scores.max(dim=1)[1]
    for i in range(len(pred)):                     # src/methods.py:96-104
        if pred[i] == 1 and y[i] == 1:
            TP += 1
        elif pred[i] == 1 and y[i] == 0:
            FP += 1
        elif pred[i] == 0 and y[i] == 1:
            FN += 1
        else:
            TN += 1
    return TP, FN, TN, FP


@pytest.mark.gpu
@pytest.mark.parametrize("B,C", [(1, 2), (63, 2), (64, 2), (1000, 2), (70001, 2), (500, 3)])
def test_confusion_kernel_follows_the_reference_rule(dev, B, C):
    g = torch.Generator().manual_seed(B)
    scores = torch.randn(B, C, generator=g)
    scores[::7] = 0.25                                # ties: .max(dim=1)[1] takes the first maximum
    y = torch.randint(0, 2, (B,), generator=g)
    if B > 10:
        y[3] = 2                                      # a label outside {0, 1} falls into the reference's else branch
    counts = torch.zeros(4, dtype=torch.int64, device=dev)
    NM.confusion_update(scores.to(dev), y.to(dev), counts)
    NM.confusion_update(scores.to(dev), y.to(dev), counts)     # accumulates across batches
    ref = _reference_loop(scores, y)
    assert counts.tolist() == [2 * v for v in ref]


@pytest.mark.gpu
def test_evaluation_function_keeps_the_reference_interface(dev, capsys):
    fx = torch.load(os.path.join(G, "rpi369_fold0.pt"), map_location="cpu", weights_only=False)

    class Data:                                       # what a PyG DataLoader yields, as far as the function looks
        def __init__(self, logp, y):
            self.logp, self.y = logp, y

        def to(self, device):
            return Data(self.logp.to(device), self.y.to(device))

    class Loader(list):
        dataset = list(range(fx["y"].numel()))

    class Model(torch.nn.Module):
        def forward(self, data):
            return data.logp

    loader = Loader(Data(fx["logp"][i:i + 50], fx["y"][i:i + 50]) for i in range(0, fx["y"].numel(), 50))
    got = NM.Accuracy_Precision_Sensitivity_Specificity_MCC(Model(), loader, dev)
    assert "TP: 42, FN: 32, TN: 51, FP: 23" in capsys.readouterr().out
    assert ["%.5f" % v for v in got] == fx["logged_metrics"]
    assert abs(NM.accuracy(Model(), loader, dev) - (42 + 51) / 148) < 1e-12
