"""Generate the committed golden fixtures from the reference's DATA artifacts (run in the build
container, where /root/reference exists):  python tests/golden/make_golden.py

Fixtures are data only -- regenerated inputs (x, edge_index, batch, y), reference checkpoint tensors
(result/<proj>/model_<k>_fold/<ep>), the reference's own logged results (metric lines of
result/<proj>/log_<k>.txt, probabilities of data/case_study/*/logs/*.txt), and the oracle's
per-layer outputs / fp64 gradients once the oracle has reproduced those logged results.

  rpi369_fold0.pt      whole RPI369 fold-0 test set (148 samples) + ckpt 50 + logged metrics
  npinter2_small.pt    24 NPInter2 (project 1223_1) fold-0 test samples + ckpt 50 conv/pool/lin weights
                       + oracle per-layer activations + fp64 conv gradients
  npinter2_katp.pt     32 fold-1 test negatives + ckpt 15 + the reference's logged P(positive)
  kat_expected.json    confusion matrices recoverable from the logs
"""
import json
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import kat, ref_conv as R  # noqa: E402


def conv_grads_fp64(x, ei, W, b, seed):
    g = torch.Generator().manual_seed(seed)
    go = torch.randn(x.size(0), W.size(1), generator=g, dtype=torch.float64)
    out, dx, dw, db = R.sage_layer_fwd_bwd(x.double(), ei, W.double(), b.double(), go)
    return {"grad_out": go.float(), "out": out.float(), "dx": dx.float(), "dw": dw.float(), "db": db.float()}


def main():
    assert kat.have_reference(), "needs /root/reference"
    # ---- RPI369, whole fold-0 test set --------------------------------------------------------
    proj = kat.Project("RPI369", "1228_1", 0)
    keys = proj.test_pos + proj.test_neg
    y = torch.tensor([1] * len(proj.test_pos) + [0] * len(proj.test_neg))
    sd = kat.load_checkpoint("1228_1", 0, 50)
    x, ei, b = proj.batch(keys)
    with torch.no_grad():
        logp, layers = R.net1_forward(sd, x, ei, b, len(keys), return_layers=True)
    cm = kat.confusion(logp, y)
    got = ["%.5f" % v for v in R.metrics_from_confusion(*cm)]
    want = kat.logged_metrics("1228_1", 0, 50)
    assert got == want, (got, want)
    torch.save({"x": x, "edge_index": ei, "batch": b, "y": y, "state_dict": sd, "logp": logp,
                "conv1_out": layers[0], "confusion_TP_FN_TN_FP": cm, "logged_metrics": want,
                "source": "RPI369.xlsx + data/*/1228_1 fold 0; result/1228_1/model_0_fold/50; result/1228_1/log_0.txt"},
               os.path.join(HERE, "rpi369_fold0.pt"))
    print("rpi369_fold0.pt", x.shape, ei.shape, cm, want)

    # ---- NPInter2 small: 24 samples, per-layer activations, gradients ------------------------------
    proj = kat.Project("NPInter2", "1223_1", 0)
    sd = kat.load_checkpoint("1223_1", 0, 50)
    # small subgraphs keep the fixture small; take 12 positives + 12 negatives with <= 120 nodes
    def small(keys_, n):
        out = []
        for k in keys_:
            if len(proj.sample(*k)[0]) <= 120:
                out.append(k)
            if len(out) == n:
                break
        return out
    kp, kn = small(proj.test_pos, 12), small(proj.test_neg, 12)
    keys = kp + kn
    y = torch.tensor([1] * len(kp) + [0] * len(kn))
    x, ei, b = proj.batch(keys)
    with torch.no_grad():
        logp, layers = R.net1_forward(sd, x, ei, b, len(keys), return_layers=True)
        agg1 = R.sage_aggregate(x, ei)
    grads = conv_grads_fp64(x, ei, sd["conv1.weight"], sd["conv1.bias"], seed=3)
    torch.save({"x": x, "edge_index": ei, "batch": b, "y": y, "state_dict": sd, "logp": logp, "agg1": agg1,
                "conv_out": layers, "conv1_grads_fp64": grads,
                "source": "NPInter2.xlsx + data/*/1223_1 fold 0 (24 test samples); result/1223_1/model_0_fold/50"},
               os.path.join(HERE, "npinter2_small.pt"))
    print("npinter2_small.pt", x.shape, ei.shape)

    # ---- KAT-P: reference's own per-sample probabilities --------------------------------------------
    proj = kat.Project("NPInter2", "1223_1", 1)
    sd = kat.load_checkpoint("1223_1", 1, 15)
    case = "1223_1_fold_1_negativeSamples_threshold_0.99"
    logged = {}
    for fn in ("case_predict_positive.txt", "case_predict_negative.txt"):
        for line in open(os.path.join(kat.REF, "data", "case_study", case, "logs", fn)):
            parts = line.rstrip("\n").split("\t")
            if len(parts) == 3:
                logged[(proj.rna_serial[parts[0]], proj.prot_serial[parts[1]])] = float(parts[2])
    keys = [k for k in proj.test_neg if k in logged and len(proj.sample(*k)[0]) <= 150][:32]
    x, ei, b = proj.batch(keys)
    with torch.no_grad():
        logp = R.net1_forward(sd, x, ei, b, len(keys))
    p_ref = torch.tensor([logged[k] for k in keys], dtype=torch.float64)
    err = float((logp[:, 1].double().exp() - p_ref).abs().max())
    assert err < 1e-5, err
    torch.save({"x": x, "edge_index": ei, "batch": b, "state_dict": sd, "p_positive_logged": p_ref,
                "source": f"data/case_study/{case}/logs (reference src/case_study_negativeSample.py:337-355); "
                          "result/1223_1/model_1_fold/15"},
               os.path.join(HERE, "npinter2_katp.pt"))
    print("npinter2_katp.pt", x.shape, "max |dP| oracle vs logged", err)

    # ---- full NPInter2 graph for BASELINE.json configs 1-3 (full-batch stacks without pooling) -------
    # N = 5,085 (4,636 ncRNA + 449 protein), 20,824 undirected = 41,648 directed edges;
    # x = [label = 1 | node2vec (fold 0) | k-mer], F = 178.  Expected outputs are the ORACLE's (no
    # reference artifact runs full-batch: parity unpinned for these stacks).
    proj = kat.Project("NPInter2", "1223_1", 0)
    pairs = torch.tensor(proj.pos + proj.neg, dtype=torch.long).t()          # [2, 20824] (rna, protein)
    ei = torch.cat([pairs, pairs.flip(0)], dim=1)                             # both directions (src/classes.py:701-704)
    assert ei.size(1) == 41648 and proj.num_nodes == 5085
    x = torch.cat([torch.ones(proj.num_nodes, 1, dtype=torch.float64), proj.feat], dim=1).float()
    sd = kat.load_checkpoint("1223_1", 0, 50)
    g = torch.Generator().manual_seed(7)

    def glorot(i, o):
        a = (6.0 / (i + o)) ** 0.5
        return (torch.rand(i, o, generator=g) * 2 - 1) * a
    gcn64 = [(glorot(178, 64), torch.zeros(64)), (glorot(64, 64), torch.zeros(64))]
    gcn256 = [(glorot(178, 256), torch.zeros(256)), (glorot(256, 256), torch.zeros(256)), (glorot(256, 256), torch.zeros(256))]
    with torch.no_grad():
        h = x
        sage_out = []
        for k in (1, 2, 3):                                                    # config 2: Net_1's convs without pooling
            h = torch.relu(R.sage_conv(h, ei, sd[f"conv{k}.weight"], sd[f"conv{k}.bias"]))
            sage_out.append(h)
        h = x
        for W, b in gcn64:                                                     # config 1
            h = torch.relu(R.gcn_conv(h, ei, W, b))
        c1 = h
        h = x
        for W, b in gcn256:                                                    # config 3 (on the NPInter2 graph)
            h = torch.relu(R.gcn_conv(h, ei, W, b))
        c3 = h
    # expected outputs are kept for a fixed sample of rows only (the heaviest node + 399 random ones)
    deg = torch.bincount(ei[1], minlength=proj.num_nodes)
    rows = torch.cat([deg.argmax().view(1), torch.randperm(proj.num_nodes, generator=g)[:399]]).sort().values
    torch.save({"x": x, "edge_index": ei.to(torch.int32), "rows": rows,
                "sage_weights": [(sd[f"conv{k}.weight"], sd[f"conv{k}.bias"]) for k in (1, 2, 3)],
                "sage3_out": sage_out[-1][rows], "sage1_out": sage_out[0][rows], "gcn64": gcn64, "gcn64_out": c1[rows],
                "gcn256": gcn256, "gcn256_out": c3[rows],
                "source": "NPInter2.xlsx + set_negativeInteractionKey_all + node2vec/k-mer of project 1223_1 fold 0; "
                          "conv weights of result/1223_1/model_0_fold/50; GCN weights seeded glorot"},
               os.path.join(HERE, "npinter2_graph.pt"))
    print("npinter2_graph.pt", x.shape, ei.shape)

    # ---- expected confusion matrices recoverable from the logs ----------------------
    table = {"1223_1": {"0/5": [1970, 113, 1922, 161], "0/25": [1980, 103, 1929, 154], "0/50": [1994, 89, 1901, 182],
                        "3/50": [1879, 203, 1993, 89], "1/50": [2033, 50, 1844, 239], "2/50": [1945, 137, 1970, 112],
                        "4/50": [2019, 63, 1888, 194]},
             "1223_1_noKmer": {"0/50": [1993, 90, 1925, 158]},
             "1228_1": {"0/50": [42, 32, 51, 23]}}
    for projname, rows in table.items():
        for k, cmv in rows.items():
            fold, ep = (int(v) for v in k.split("/"))
            want = kat.logged_metrics(projname, fold, ep)
            got = ["%.5f" % v for v in R.metrics_from_confusion(*cmv)]
            assert got == want, (projname, k, got, want)
            rows[k] = {"TP_FN_TN_FP": cmv, "logged": want}
    json.dump(table, open(os.path.join(HERE, "kat_expected.json"), "w"), indent=1)
    print("kat_expected.json ok")


if __name__ == "__main__":
    main()
