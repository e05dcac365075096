"""Generate tests/golden/npinter2_kat_full.pt (run in the build container, where /root/reference exists):
    python tests/golden/make_npinter2_kat_full.py

The REST of the reference-pinned known-answer tests of SURVEY.md 8(c), as DATA for the GPU box.
tests/golden/npinter2_folds.pt already carries the inputs of all five folds of project 1223_1 (pairs, k-mer block, every
fold's node2vec block and test keys), the fold-0 epoch-50 checkpoint and the fold-1 case-study probabilities; this file
adds everything else the reference logged for that project and for project 1227_1:

  checkpoints        result/1223_1/model_{1,2,3,4}_fold/50, model_0_fold/{5,25}, result/1223_1_noKmer/model_0_fold/50
                     (the F = 65 variant: features [label | node2vec], src/generate_dataset.py:263-267),
                     result/1223_1/model_{2,3,4}_fold/15, result/1227_1/model_{0,1}_fold/20
  confusion matrices the (TP, FN, TN, FP) each logged metric line of result/<project>/log_<k>.txt implies
  probabilities      data/case_study/1223_1_fold_{2,3,4}_negativeSamples_threshold_0.99/logs and
                     data/case_study/1227_1_fold_{0,1}_negativeSamples_threshold_0.95/logs: P(positive) of every test-fold
                     negative (src/case_study_negativeSample.py:337-355)
  project 1227_1     its own pair list (another draw of negatives), node2vec blocks and test keys of folds 0 and 1

Before anything is written the CPU oracle (oracle/kat.py + oracle/ref_conv.py) must reproduce every one of them in this
container: each confusion matrix exactly, each probability to 1e-5.  Only data is stored: no reference source text.
"""
import os
import sys
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import kat, ref_conv as R  # noqa: E402

CONFUSION = [                     # (result project, fold, epoch, no_kmer)
    ("1223_1", 0, 5, False), ("1223_1", 0, 25, False),
    ("1223_1", 1, 50, False), ("1223_1", 2, 50, False), ("1223_1", 3, 50, False), ("1223_1", 4, 50, False),
    ("1223_1_noKmer", 0, 50, True),
]
PROBS = [                         # (data project, fold, epoch, case directory)
    ("1223_1", 2, 15, "1223_1_fold_2_negativeSamples_threshold_0.99"),
    ("1223_1", 3, 15, "1223_1_fold_3_negativeSamples_threshold_0.99"),
    ("1223_1", 4, 15, "1223_1_fold_4_negativeSamples_threshold_0.99"),
    ("1227_1", 0, 20, "1227_1_fold_0_negativeSamples_threshold_0.95"),
    ("1227_1", 1, 20, "1227_1_fold_1_negativeSamples_threshold_0.95"),
]


def logged_probs(proj, case):
    logged = {}
    for fn in ("case_predict_positive.txt", "case_predict_negative.txt"):
        for line in open(os.path.join(kat.REF, "data", "case_study", case, "logs", fn)):
            parts = line.rstrip("\n").split("\t")
            if len(parts) == 3:
                logged[(proj.rna_serial[parts[0]], proj.prot_serial[parts[1]])] = float(parts[2])
    return logged


def main():
    assert kat.have_reference(), "needs /root/reference"
    out = {"confusion": [], "probabilities": [], "projects": {}}
    projects = {}

    def project(name, fold, no_kmer=False):
        key = (name, fold, no_kmer)
        if key not in projects:
            projects[key] = kat.Project("NPInter2", name, fold, no_kmer)
        return projects[key]

    for rp, fold, epoch, no_kmer in CONFUSION:
        t = time.time()
        proj = project("1223_1", fold, no_kmer)
        sd = kat.load_checkpoint(rp, fold, epoch)
        keys = proj.test_pos + proj.test_neg
        y = torch.tensor([1] * len(proj.test_pos) + [0] * len(proj.test_neg))
        cm = kat.confusion(kat.predict(proj, keys, sd), y)
        want = kat.logged_metrics(rp, fold, epoch)
        got = ["%.5f" % v for v in R.metrics_from_confusion(*cm)]
        assert got == want, (rp, fold, epoch, cm, got, want)
        out["confusion"].append({"project": "1223_1", "result_project": rp, "fold": fold, "epoch": epoch, "no_kmer": no_kmer,
                                 "state_dict": sd, "TP_FN_TN_FP": list(cm), "logged_metrics": want})
        print(f"confusion {rp} fold {fold} epoch {epoch}: {cm} = {want}  ({time.time() - t:.0f} s)", flush=True)

    for name, fold, epoch, case in PROBS:
        t = time.time()
        proj = project(name, fold)
        logged = logged_probs(proj, case)
        assert all(k in logged for k in proj.test_neg), (case, len(logged), len(proj.test_neg))
        sd = kat.load_checkpoint(name, fold, epoch)
        p_ref = torch.tensor([logged[k] for k in proj.test_neg], dtype=torch.float64)
        p = kat.predict(proj, proj.test_neg, sd)[:, 1].double().exp()
        err = float((p - p_ref).abs().max())
        assert err <= 1e-5, (case, err)
        out["probabilities"].append({"project": name, "fold": fold, "epoch": epoch, "case": case, "state_dict": sd,
                                     "p_positive_logged": p_ref, "oracle_max_abs_err": err})
        print(f"probabilities {case}: {p_ref.numel()} samples, oracle max |dP| {err:.2e}  ({time.time() - t:.0f} s)", flush=True)

    # project 1227_1: its own negatives, embeddings and test keys (1223_1's travel in npinter2_folds.pt)
    p0, p1 = project("1227_1", 0), project("1227_1", 1)
    ref = project("1223_1", 2)
    assert p0.pos == p1.pos == ref.pos and p0.neg == p1.neg and p0.neg != ref.neg and p0.num_nodes == ref.num_nodes
    assert torch.equal(p0.feat[:, 64:], ref.feat[:, 64:])                       # the k-mer block is the dataset's, not the project's
    out["projects"]["1227_1"] = {
        "pairs": torch.tensor(p0.pos + p0.neg, dtype=torch.int32),
        "label": torch.tensor([1] * len(p0.pos) + [0] * len(p0.neg), dtype=torch.uint8),
        "num_nodes": p0.num_nodes,
        "folds": {k: {"node2vec": p.feat[:, :64].to(torch.float32),
                      "test_pos": torch.tensor(p.test_pos, dtype=torch.int32),
                      "test_neg": torch.tensor(p.test_neg, dtype=torch.int32)} for k, p in ((0, p0), (1, p1))}}
    out["source"] = ("result/1223_1/model_{0..4}_fold/{5,15,25,50}, result/1223_1_noKmer/model_0_fold/50, result/1227_1/"
                     "model_{0,1}_fold/20 and the log_k.txt metric lines; data/case_study/1223_1_fold_{2,3,4}_..._0.99/logs, "
                     "1227_1_fold_{0,1}_..._0.95/logs; data/set_allInteractionKey/1227_1, data/node2vec_result/1227_1/"
                     "training_{0,1}; pairs / k-mers / node2vec of 1223_1: tests/golden/npinter2_folds.pt")
    path = os.path.join(HERE, "npinter2_kat_full.pt")
    torch.save(out, path)
    print("npinter2_kat_full.pt", os.path.getsize(path), "bytes;", len(out["confusion"]), "confusion matrices,",
          len(out["probabilities"]), "probability sets")


if __name__ == "__main__":
    main()
