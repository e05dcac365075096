"""Generate tests/golden/npinter2_folds.pt (run in the build container, where /root/reference exists):
    python tests/golden/make_npinter2_folds.py

The whole NPInter2 workload of reference project 1223_1 as DATA, so that the GPU box can run the reference's real
regime end to end:

  shared   pairs [20824, 2] (positives in xlsx order, then set_negativeInteractionKey_all), k-mer block [5085, 113]
  fold 0   node2vec block (training_0), test keys (2,083 + 2,083), checkpoint result/1223_1/model_0_fold/50 and the
           confusion matrix its logged metric line implies (TP 1994 FN 89 TN 1901 FP 182, result/1223_1/log_0.txt);
           the reference's 5-fold test-accuracy range at epoch 50 (log_0..4.txt)
  fold 1   node2vec block (training_1), test keys, checkpoint model_1_fold/15 and the 2,083 per-sample P(positive)
           the reference logged for the fold's test negatives
           (data/case_study/1223_1_fold_1_negativeSamples_threshold_0.99/logs, src/case_study_negativeSample.py:337-355)
  folds 2-4  node2vec blocks, test keys and the logged epoch-50 test metrics (result/1223_1/log_k.txt): the inputs of the
           reference's five-fold cross-validation, for examples/train_npinter2.py --fold all

Before anything is written the CPU oracle (oracle/kat.py + oracle/ref_conv.py) must reproduce both: the confusion
matrix exactly and every probability to 1e-5.  Only data is stored: no reference source text.
"""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import kat, ref_conv as R  # noqa: E402


def fold_block(proj):
    emb = proj.feat[:, :64].to(torch.float32)
    return {"node2vec": emb,
            "test_pos": torch.tensor(proj.test_pos, dtype=torch.int32), "test_neg": torch.tensor(proj.test_neg, dtype=torch.int32)}


def main():
    assert kat.have_reference(), "needs /root/reference"
    p0 = kat.Project("NPInter2", "1223_1", 0)
    p1 = kat.Project("NPInter2", "1223_1", 1)
    assert p0.pos == p1.pos and p0.neg == p1.neg and torch.equal(p0.feat[:, 64:], p1.feat[:, 64:])
    pairs = torch.tensor(p0.pos + p0.neg, dtype=torch.int32)
    label = torch.tensor([1] * len(p0.pos) + [0] * len(p0.neg), dtype=torch.uint8)
    kmer = p0.feat[:, 64:].to(torch.float32)
    # (the oracle builds x in float64 and casts the finished batch to float32; rounding each stored value first gives
    # the same float32 rows: the cast is elementwise)
    out = {"pairs": pairs, "label": label, "kmer": kmer, "num_nodes": p0.num_nodes}

    # fold 0: whole-fold KAT
    sd0 = kat.load_checkpoint("1223_1", 0, 50)
    keys = p0.test_pos + p0.test_neg
    y = torch.tensor([1] * len(p0.test_pos) + [0] * len(p0.test_neg))
    cm = kat.confusion(kat.predict(p0, keys, sd0), y)
    want = kat.logged_metrics("1223_1", 0, 50)
    assert ["%.5f" % v for v in R.metrics_from_confusion(*cm)] == want and cm == (1994, 89, 1901, 182), (cm, want)
    accs = [float(kat.logged_metrics("1223_1", k, 50)[0]) for k in range(5)]
    f0 = fold_block(p0)
    f0.update(state_dict=sd0, confusion_TP_FN_TN_FP=list(cm), logged_metrics=want, logged_test_acc_5fold_epoch50=accs,
              logged_wall_seconds=1413.46199965477)
    out["fold0"] = f0

    # fold 1: KAT-P
    case = "1223_1_fold_1_negativeSamples_threshold_0.99"
    sd1 = kat.load_checkpoint("1223_1", 1, 15)
    logged = {}
    for fn in ("case_predict_positive.txt", "case_predict_negative.txt"):
        for line in open(os.path.join(kat.REF, "data", "case_study", case, "logs", fn)):
            parts = line.rstrip("\n").split("\t")
            if len(parts) == 3:
                logged[(p1.rna_serial[parts[0]], p1.prot_serial[parts[1]])] = float(parts[2])
    assert all(k in logged for k in p1.test_neg) and len(p1.test_neg) == 2083
    p_ref = torch.tensor([logged[k] for k in p1.test_neg], dtype=torch.float64)
    p = kat.predict(p1, p1.test_neg, sd1)[:, 1].double().exp()
    err = float((p - p_ref).abs().max())
    assert err <= 1e-5, err
    f1 = fold_block(p1)
    f1.update(state_dict=sd1, p_positive_logged=p_ref, case=case)
    out["fold1"] = f1
    # folds 2-4: inputs and the reference's logged epoch-50 test metrics only (five-fold cross-validation of the example)
    for k in (2, 3, 4):
        pk = kat.Project("NPInter2", "1223_1", k)
        assert pk.pos == p0.pos and pk.neg == p0.neg
        fk = fold_block(pk)
        fk.update(logged_metrics=kat.logged_metrics("1223_1", k, 50))
        out[f"fold{k}"] = fk
    out["fold1"]["logged_metrics"] = kat.logged_metrics("1223_1", 1, 50)
    out["source"] = ("NPInter2.xlsx, data/set_allInteractionKey/1223_1, data/node2vec_result/1223_1/training_{0,1}, "
                     "data/lncRNA_3_mer + protein_2_mer; result/1223_1/model_0_fold/50, model_1_fold/15, log_*.txt; "
                     f"data/case_study/{case}/logs")
    path = os.path.join(HERE, "npinter2_folds.pt")
    torch.save(out, path)
    print("npinter2_folds.pt", os.path.getsize(path), "bytes; fold 0 confusion", cm, "fold 1 max |dP|", err, "5-fold acc", accs)


if __name__ == "__main__":
    main()
