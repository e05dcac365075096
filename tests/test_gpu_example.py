"""The end-to-end training example (examples/train_rpi369.py): target pairs -> device-built batches -> Net_1 ->
loss -> backward -> Adam, then evaluation through the confusion kernel, on the RPI369 vectors."""
import importlib.util
import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_training_loop_learns_on_rpi369(dev):
    spec = importlib.util.spec_from_file_location("train_rpi369", os.path.join(ROOT, "examples", "train_rpi369.py"))
    ex = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ex)
    torch.manual_seed(0)
    ig, train_keys, train_y, test_keys, test_y, F_in = ex.load_project(dev)
    assert train_keys.size(0) == 590 and test_keys.size(0) == 148 and F_in == 178
    assert int(test_y.sum()) == 74 and int(train_y.sum()) == 295
    model = ex.Net_1(F_in).to(dev)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-3)
    loader = ex.KeyLoader(ig, train_keys, train_y, 200, shuffle=True, seed=0)
    losses = []
    for epoch in range(12):
        model.train()
        tot = 0.0
        for data in loader:
            opt.zero_grad()
            loss = F.nll_loss(model(data), data.y)
            loss.backward()
            opt.step()
            tot += data.num_graphs * float(loss.detach())
        losses.append(tot / train_keys.size(0))
    assert all(l == l for l in losses)                       # no NaN
    assert losses[-1] < losses[0] - 0.03, losses             # it learns
    from npi_gnn_amd import metrics as NM
    m = NM.Accuracy_Precision_Sensitivity_Specificity_MCC(model, ex.KeyLoader(ig, test_keys, test_y, 200, shuffle=False), dev)
    assert all(0.0 <= v <= 1.0 for v in m[:4]) and -1.0 <= m[4] <= 1.0
