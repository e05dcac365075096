"""The end-to-end training flow (npi_gnn_amd.net1 + examples/): target pairs -> device-built batches -> Net_1 -> loss ->
backward -> Adam, evaluation through the confusion kernel -- the reference's loop (src/train_with_twoDataset.PY:46-57,
142-184) on the RPI369 vectors."""
import importlib.util
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _example(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "examples", name + ".py"))
    ex = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ex)
    return ex


def test_training_loop_learns_on_rpi369(dev):
    from npi_gnn_amd import net1
    ex = _example("train_rpi369")
    torch.manual_seed(0)
    ig, train_keys, train_y, test_keys, test_y, F_in = ex.load_project(dev)
    assert train_keys.size(0) == 590 and test_keys.size(0) == 148 and F_in == 178
    assert int(test_y.sum()) == 74 and int(train_y.sum()) == 295
    g = torch.Generator().manual_seed(0)
    train_loader = net1.KeyLoader(ig, train_keys, train_y, 200).shuffle(g)
    test_loader = net1.KeyLoader(ig, test_keys, test_y, 200)
    # the loader does not reshuffle: two passes yield the same batches (reference: DataLoader without shuffle=)
    a = [b.y.clone() for b in train_loader]
    b = [b.y.clone() for b in train_loader]
    assert len(a) == 3 and all(torch.equal(u, v) for u, v in zip(a, b))
    model = net1.Net_1(F_in).to(dev)
    lines = []
    res = net1.fit(model, train_loader, test_loader, dev, num_of_epoch=12, log=lines.append)
    losses = res["loss"]
    assert all(l == l for l in losses)                       # no NaN
    assert losses[-1] < losses[0] - 0.03, losses             # it learns
    # the scheduler is stepped exactly in the epochs whose loss rose (src/train_with_twoDataset.PY:158-160)
    assert res["lr_steps"] == sum(1 for i in range(1, len(losses)) if losses[i] > losses[i - 1])
    # metrics every 5th epoch on both loaders, and once at the end (:163-172, :186-193)
    assert [l.split(",")[0] + "," + l.split(",")[1] for l in lines if "dataset" in l] == [
        "Epoch: 005, training dataset", "Epoch: 005, testing dataset", "Epoch: 010, training dataset",
        "Epoch: 010, testing dataset", "result, training dataset", "result, testing dataset"]
    m = res["test"]
    assert all(0.0 <= v <= 1.0 for v in m[:4]) and -1.0 <= m[4] <= 1.0


def test_training_step_replayed_from_hip_graphs_equals_the_eager_loop(dev):
    """net1.GraphedEpoch: every batch's whole step -- Net_1 forward with its three pooling layers, loss,
    backward, Adam -- captured once and replayed; with dropout off the loss curve must be the eager loop's."""
    from npi_gnn_amd import net1
    ex = _example("train_rpi369")
    ig, train_keys, train_y, test_keys, test_y, F_in = ex.load_project(dev)
    g = torch.Generator().manual_seed(0)
    loader = net1.KeyLoader(ig, train_keys, train_y, 200).shuffle(g)
    curves = []
    for capture_after in (1, 10 ** 9):
        torch.manual_seed(0)
        model = net1.Net_1(F_in, dropout=0.0).to(dev)
        opt = torch.optim.Adam(model.parameters(), lr=torch.tensor(1e-3, device=dev), weight_decay=1e-3, capturable=True)
        epoch = net1.GraphedEpoch(model, loader, opt, dev, capture_after=capture_after)
        curves.append([epoch() for _ in range(6)])
        if capture_after == 1:
            assert all(gr is not None for gr in epoch.graphs)
    a, b = curves
    assert all(abs(u - v) <= 1e-5 * max(1.0, abs(v)) for u, v in zip(a, b)), (a, b)
    assert a[-1] < a[0]
    # and through fit(): the reference's loop with captured steps, scheduler included
    torch.manual_seed(0)
    model = net1.Net_1(F_in).to(dev)
    res = net1.fit(model, loader, net1.KeyLoader(ig, test_keys, test_y, 200), dev, num_of_epoch=8, log=lambda s: None,
                   capture=True)
    assert res["loss"][-1] < res["loss"][0] and all(l == l for l in res["loss"])
