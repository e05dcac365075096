"""The reference's REAL workload on the MI355X: project 1223_1 (NPInter2) from
tests/golden/npinter2_folds.pt -- made by tests/golden/make_npinter2_folds.py from the reference's data files after
the CPU oracle had reproduced the same numbers.

  fold 0  4,166 test keys -> device extraction -> Net_1 (reference checkpoint, loaded with load_state_dict) ->
          confusion kernel  ==  TP 1994 / FN 89 / TN 1901 / FP 182 and the metric line of result/1223_1/log_0.txt
  fold 1  all 2,083 per-sample P(positive) the reference logged for the fold's test negatives, <= 1e-5
  train   a few epochs of the reference's loop on the 16,658 training pairs: the loss falls, accuracy is sane
"""
import importlib.util
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def ex():
    spec = importlib.util.spec_from_file_location("train_npinter2", os.path.join(ROOT, "examples", "train_npinter2.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_whole_fold_kat_through_device_extraction(dev, ex):
    from npi_gnn_amd import metrics as NM, net1
    ig, train_keys, train_y, test_keys, test_y, F_in, fx = ex.load_fold(dev, 0)
    assert (train_keys.size(0), test_keys.size(0), F_in) == (16658, 4166, 178)
    model = net1.Net_1(F_in).to(dev)
    model.load_state_dict(fx["fold0"]["state_dict"])          # the reference's checkpoint, unchanged (src/test.py:41)
    model.eval()
    counts = torch.zeros(4, dtype=torch.int64, device=dev)
    with torch.no_grad():
        for data in net1.KeyLoader(ig, test_keys, test_y, 200):
            NM.confusion_update(model(data), data.y, counts)
    assert counts.tolist() == fx["fold0"]["confusion_TP_FN_TN_FP"] == [1994, 89, 1901, 182]
    m = NM.Accuracy_Precision_Sensitivity_Specificity_MCC(model, net1.KeyLoader(ig, test_keys, test_y, 200), dev)
    assert ["%.5f" % v for v in m] == fx["fold0"]["logged_metrics"]
    # batching does not change a sample's prediction: batch size 1 (the case-study scripts) on a slice, 64 on the rest
    with torch.no_grad():
        a = torch.cat([model(d) for d in net1.KeyLoader(ig, test_keys[:48], test_y[:48], 1)])
        b = torch.cat([model(d) for d in net1.KeyLoader(ig, test_keys[:48], test_y[:48], 64)])
    assert torch.allclose(a, b, atol=1e-5, rtol=1e-5)


def test_all_case_study_probabilities_fold1(dev, ex):
    from npi_gnn_amd import net1
    ig, _, _, _, _, F_in, fx = ex.load_fold(dev, 1)
    f1 = fx["fold1"]
    keys = f1["test_neg"].long().to(dev)
    model = net1.Net_1(F_in).to(dev)
    model.load_state_dict(f1["state_dict"])
    model.eval()
    y = torch.zeros(keys.size(0), dtype=torch.long, device=dev)
    with torch.no_grad():
        logp = torch.cat([model(d) for d in net1.KeyLoader(ig, keys, y, 200)])
    p = logp[:, 1].double().exp().cpu()
    ref = f1["p_positive_logged"]
    assert p.numel() == ref.numel() == 2083
    assert float((p - ref).abs().max()) <= 1e-5              # observed on the CPU oracle: 6.7e-7


def test_reference_training_loop_on_the_real_fold(dev, ex):
    from npi_gnn_amd import net1
    torch.manual_seed(0)
    ig, train_keys, train_y, test_keys, test_y, F_in, fx = ex.load_fold(dev, 0)
    g = torch.Generator().manual_seed(0)
    train_loader = net1.KeyLoader(ig, train_keys, train_y, 200).shuffle(g)
    test_loader = net1.KeyLoader(ig, test_keys, test_y, 200).shuffle(g)
    assert len(train_loader) == 84                            # 84 batches per epoch
    model = net1.Net_1(F_in).to(dev)
    res = net1.fit(model, train_loader, test_loader, dev, num_of_epoch=3, log=lambda s: None, eval_train=False)
    assert res["loss"][-1] < res["loss"][0]
    assert res["test"][0] > 0.85                              # reference: 0.934 at epoch 5 (log_0.txt)


def test_every_fold_of_the_cross_validation_is_loadable(dev, ex):
    """The fixture carries the inputs of all five folds (examples/train_npinter2.py --fold all): 4/5 of the 20,824 pairs train,
    1/5 test, the test keys never usable as context (src/generate_dataset.py:296-299), logged reference accuracy per fold."""
    seen = set()
    for fold in range(5):
        ig, train_keys, train_y, test_keys, test_y, F_in, fx = ex.load_fold(dev, fold)
        assert F_in == 178 and train_keys.size(0) + test_keys.size(0) == 20824
        assert test_keys.size(0) in (4166, 4164) and int(test_y.sum()) * 2 == test_keys.size(0)
        codes = set((test_keys[:, 0] * 5085 + test_keys[:, 1]).tolist())
        assert not (codes & seen)                                   # the folds' test sets are disjoint
        seen |= codes
        assert 0.92 < float(fx[f"fold{fold}"]["logged_metrics"][0]) < 0.95
    assert len(seen) == 20824


# ---- the REST of the reference-pinned KAT on the HIP path (tests/golden/npinter2_kat_full.pt) ----------------
@pytest.fixture(scope="module")
def full():
    return torch.load(os.path.join(ROOT, "tests", "golden", "npinter2_kat_full.pt"), map_location="cpu", weights_only=False)


def _interaction_graph(dev, pairs, num_nodes, fold_block, kmer):
    """(InteractionGraph, F_in) of one fold: the fold's test keys are never usable as context (src/generate_dataset.py:296-299);
    ``kmer`` None: the noKmer variant, features [label | node2vec] (src/generate_dataset.py:263-267)"""
    from npi_gnn_amd.subgraph import InteractionGraph
    pairs = pairs.long()
    test = torch.cat([fold_block["test_pos"], fold_block["test_neg"]]).long()
    usable = ~torch.isin(pairs[:, 0] * num_nodes + pairs[:, 1], test[:, 0] * num_nodes + test[:, 1])
    feat = fold_block["node2vec"] if kmer is None else torch.cat([fold_block["node2vec"], kmer], dim=1)
    return InteractionGraph(pairs.to(dev), usable.to(dev), feat.contiguous().to(dev), num_nodes=num_nodes), feat.size(1) + 1


def _predict(dev, ig, F_in, sd, keys, batch=200):
    from npi_gnn_amd import net1
    model = net1.Net_1(F_in).to(dev)
    model.load_state_dict(sd)                                   # the reference's checkpoint, unchanged (src/test.py:41)
    model.eval()
    y = torch.zeros(keys.size(0), dtype=torch.long, device=dev)
    with torch.no_grad():
        return torch.cat([model(d) for d in net1.KeyLoader(ig, keys, y, batch)])


def test_every_logged_confusion_matrix_of_the_project_on_the_hip_path(dev, full):
    """All the (fold, epoch) pairs of project 1223_1 whose metric line SURVEY 8(c) lists -- folds 1-4 at epoch 50, fold 0 at
    epochs 5 and 25 (epoch 50: test_whole_fold_kat_through_device_extraction) -- and the noKmer variant (F = 65: a 260-byte row
    pitch through subgraph.hip, the scalar-lane aggregation kernel and the guarded GEMM): test keys -> device extraction ->
    Net_1 with the reference's checkpoint -> confusion kernel == the matrix the reference's logged line implies, and the five
    %.5f metrics of result/<project>/log_<k>.txt."""
    from npi_gnn_amd import metrics as NM, net1
    fx = torch.load(os.path.join(ROOT, "tests", "golden", "npinter2_folds.pt"), map_location="cpu", weights_only=False)
    assert len(full["confusion"]) == 7
    seen = set()
    for case in full["confusion"]:
        fb = fx[f"fold{case['fold']}"]
        ig, F_in = _interaction_graph(dev, fx["pairs"], fx["num_nodes"], fb, None if case["no_kmer"] else fx["kmer"])
        assert F_in == (65 if case["no_kmer"] else 178)
        keys = torch.cat([fb["test_pos"], fb["test_neg"]]).long().to(dev)
        y = torch.cat([torch.ones(fb["test_pos"].size(0)), torch.zeros(fb["test_neg"].size(0))]).long().to(dev)
        model = net1.Net_1(F_in).to(dev)
        model.load_state_dict(case["state_dict"])
        model.eval()
        counts = torch.zeros(4, dtype=torch.int64, device=dev)
        with torch.no_grad():
            for data in net1.KeyLoader(ig, keys, y, 200):
                NM.confusion_update(model(data), data.y, counts)
        tag = (case["result_project"], case["fold"], case["epoch"])
        assert counts.tolist() == case["TP_FN_TN_FP"], (tag, counts.tolist(), case["TP_FN_TN_FP"])
        m = NM.Accuracy_Precision_Sensitivity_Specificity_MCC(model, net1.KeyLoader(ig, keys, y, 200), dev)
        assert ["%.5f" % v for v in m] == case["logged_metrics"], tag
        seen.add(tag)
        del ig, model
    assert seen == {("1223_1", 0, 5), ("1223_1", 0, 25), ("1223_1", 1, 50), ("1223_1", 2, 50), ("1223_1", 3, 50), ("1223_1", 4, 50),
                    ("1223_1_noKmer", 0, 50)}


def test_every_logged_case_study_probability_on_the_hip_path(dev, full):
    """KAT-P beyond fold 1: P(positive) of every test-fold negative as the reference logged it
    (src/case_study_negativeSample.py:337-355) -- project 1223_1 folds 2, 3, 4 (checkpoint 15) and project 1227_1 folds 0, 1
    (checkpoint 20; its own draw of negatives, embeddings and test keys) -- 10,412 samples in all, each <= 1e-5."""
    fx = torch.load(os.path.join(ROOT, "tests", "golden", "npinter2_folds.pt"), map_location="cpu", weights_only=False)
    total = 0
    assert len(full["probabilities"]) == 5
    for case in full["probabilities"]:
        if case["project"] == "1223_1":
            pairs, N, fb = fx["pairs"], fx["num_nodes"], fx[f"fold{case['fold']}"]
        else:
            pr = full["projects"][case["project"]]
            pairs, N, fb = pr["pairs"], pr["num_nodes"], pr["folds"][case["fold"]]
        ig, F_in = _interaction_graph(dev, pairs, N, fb, fx["kmer"])
        keys = fb["test_neg"].long().to(dev)
        p = _predict(dev, ig, F_in, case["state_dict"], keys)[:, 1].double().exp().cpu()
        ref = case["p_positive_logged"]
        assert p.numel() == ref.numel() >= 2082
        err = float((p - ref).abs().max())
        assert err <= 1e-5, (case["case"], err)                  # the CPU oracle: case["oracle_max_abs_err"] (<= 1.7e-6)
        total += p.numel()
        del ig
    assert total == 2082 * 3 + 2083 * 2
