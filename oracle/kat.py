"""Known-answer test that PINS the oracle -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

The reference has no tests, but it ships trained checkpoints, the logged test metrics of every
checkpoint, per-sample predicted probabilities of case studies, and every input needed to rebuild
the test sets.  This module rebuilds a project's fold test set from those inputs (restating the
reference's sample-construction rules, cited inline), runs ``oracle.ref_conv.net1_forward`` with a
reference checkpoint, and compares with the reference's own logged results:

  KAT    confusion matrix / 5 metrics   vs  result/<proj>/log_<k>.txt          (exact, %.5f)
  KAT-P  P(positive) per test negative  vs  data/case_study/<case>/logs/*.txt  (<= 1e-5)

It reads ``/root/reference`` (or $NPI_REFERENCE) and therefore runs only in the build container;
``tests/golden/make_golden.py`` uses it to emit the small committed fixtures that travel.
No reference source is imported or copied: only its DATA files are read.
"""
from __future__ import annotations

import math
import os
import re
import zipfile
from typing import Dict, List, Set, Tuple

import torch

from . import ref_conv as R

REF = os.environ.get("NPI_REFERENCE", "/root/reference")


def have_reference() -> bool:
    return os.path.isdir(os.path.join(REF, "data", "source_database_data"))


# --------------------------------------------------------------------------------------
# inputs
# --------------------------------------------------------------------------------------
def read_xlsx_rows(path: str) -> List[List[str]]:
    """First sheet of an .xlsx as rows of strings, without openpyxl (an .xlsx is a zip of XML)."""
    z = zipfile.ZipFile(path)
    shared: List[str] = []
    if "xl/sharedStrings.xml" in z.namelist():
        s = z.read("xl/sharedStrings.xml").decode("utf8")
        for si in re.findall(r"<si>(.*?)</si>", s, flags=re.S):
            shared.append("".join(re.findall(r"<t[^>]*>(.*?)</t>", si, flags=re.S)))
    sheet_name = sorted(n for n in z.namelist() if re.match(r"xl/worksheets/sheet\d+\.xml", n))[0]
    s = z.read(sheet_name).decode("utf8")
    rows = []
    for row in re.findall(r"<row[^>]*>(.*?)</row>", s, flags=re.S):
        vals = []
        for attrs, body in re.findall(r"<c([^>]*?)(?:/>|>(.*?)</c>)", row, flags=re.S):
            t = re.search(r'\bt="(\w+)"', attrs)
            v = re.search(r"<v>(.*?)</v>", body or "", flags=re.S)
            if t and t.group(1) == "s":
                vals.append(shared[int(v.group(1))])
            elif t and t.group(1) == "inlineStr":
                vals.append("".join(re.findall(r"<t[^>]*>(.*?)</t>", body or "", flags=re.S)))
            else:
                vals.append(v.group(1) if v else "")
        rows.append(vals)
    return rows


def _unescape(s: str) -> str:
    return s.replace("&amp;", "&").replace("&lt;", "<").replace("&gt;", ">").replace("&quot;", '"').replace("&apos;", "'")


def read_key_set(path: str) -> List[Tuple[int, int]]:
    out = []
    for line in open(path):
        line = line.strip().replace("(", "").replace(")", "").replace(" ", "")
        if line:
            a, b = line.split(",")[:2]
            out.append((int(a), int(b)))
    return out


class Project:
    """Whole-graph state of one reference project/fold (reference src/generate_dataset.py:224-305)."""

    def __init__(self, dataset: str, project: str, fold: int, no_kmer: bool = False):
        self.dataset, self.project, self.fold, self.no_kmer = dataset, project, fold, no_kmer
        rows = read_xlsx_rows(os.path.join(REF, "data", "source_database_data", dataset + ".xlsx"))[1:]
        # serial numbers: ONE counter, RNA first then protein per row (src/generate_edgelist.py:70-85)
        self.rna_serial: Dict[str, int] = {}
        self.prot_serial: Dict[str, int] = {}
        self.name: Dict[int, str] = {}
        self.pos: List[Tuple[int, int]] = []
        self.neg: List[Tuple[int, int]] = []
        n = 0
        for r in rows:
            if len(r) < 3 or r[0] == "":
                continue
            rna, prot, label = _unescape(r[0]), _unescape(r[1]), int(float(r[2]))
            if rna not in self.rna_serial:
                self.rna_serial[rna] = n
                self.name[n] = rna
                n += 1
            if prot not in self.prot_serial:
                self.prot_serial[prot] = n
                self.name[n] = prot
                n += 1
            key = (self.rna_serial[rna], self.prot_serial[prot])
            (self.pos if label == 1 else self.neg).append(key)
        self.num_nodes = n
        kdir = os.path.join(REF, "data", "set_allInteractionKey", project)
        neg_all = os.path.join(kdir, "set_negativeInteractionKey_all")
        if not self.neg and os.path.exists(neg_all):      # balanced datasets: negatives were sampled
            self.neg = read_key_set(neg_all)              # (src/generate_dataset.py:236-242)
        self.test_pos = read_key_set(os.path.join(kdir, f"set_interactionKey_test_{fold}"))
        self.test_neg = read_key_set(os.path.join(kdir, f"set_negativeInteractionKey_test_{fold}"))
        self.cannot: Set[Tuple[int, int]] = set(self.test_pos) | set(self.test_neg)   # :296-299
        # adjacency in interaction_list order: positives in file order, then negatives
        self.rna_adj: Dict[int, List[int]] = {}
        self.prot_adj: Dict[int, List[int]] = {}
        for l, p in self.pos + self.neg:
            self.rna_adj.setdefault(l, []).append(p)
            self.prot_adj.setdefault(p, []).append(l)
        self.feat = self._features()

    def _features(self) -> torch.Tensor:
        """[num_nodes, 64 (+113)]: node2vec | k-mer (src/generate_dataset.py:55-75, 87-119)."""
        n = self.num_nodes
        emb = torch.zeros(n, 64, dtype=torch.float64)
        path = os.path.join(REF, "data", "node2vec_result", self.project, f"training_{self.fold}", "result.emb")
        with open(path) as f:
            f.readline()
            for line in f:
                arr = line.strip().split(" ")
                if len(arr) == 65:
                    emb[int(arr[0])] = torch.tensor([float(v) for v in arr[1:]], dtype=torch.float64)
        if self.no_kmer:
            return emb
        kmer = torch.zeros(n, 113, dtype=torch.float64)

        def load(path, serial, lo, width):
            seen = set()
            lines = open(path).read().split("\n")
            for i, line in enumerate(lines):
                if line.startswith(">"):
                    nm = line.strip()[1:]
                    if nm in serial and nm not in seen:       # first occurrence wins
                        seen.add(nm)
                        vals = lines[i + 1].strip().split("\t")
                        assert len(vals) == width, (nm, len(vals))
                        kmer[serial[nm], lo:lo + width] = torch.tensor([float(v) for v in vals], dtype=torch.float64)
        load(os.path.join(REF, "data", "lncRNA_3_mer", self.dataset, "lncRNA_3_mer.txt"), self.rna_serial, 0, 64)
        load(os.path.join(REF, "data", "protein_2_mer", self.dataset, "protein_2_mer.txt"), self.prot_serial, 64, 49)
        return torch.cat([emb, kmer], dim=1)

    # 1-hop enclosing subgraph of (l, p)  (reference src/classes.py:652-733)
    def sample(self, l: int, p: int):
        ids = {l: 0, p: 1}
        order = [l, p]
        edges = {(l, p)}
        for q in self.rna_adj.get(l, []):
            if (l, q) not in self.cannot:
                edges.add((l, q))
                if q not in ids:
                    ids[q] = len(order)
                    order.append(q)
        for m in self.prot_adj.get(p, []):
            if (m, p) not in self.cannot:
                edges.add((m, p))
                if m not in ids:
                    ids[m] = len(order)
                    order.append(m)
        src, dst = [], []
        for a, b in edges:
            ia, ib = ids[a], ids[b]
            src += [ia, ib]
            dst += [ib, ia]
        return order, src, dst

    def batch(self, keys: List[Tuple[int, int]]):
        """PyG Batch collate of the samples of `keys`: x [n,F] fp32, edge_index int64, batch."""
        node_ids, labels, src, dst, bvec = [], [], [], [], []
        off = 0
        for g, (l, p) in enumerate(keys):
            order, s, d = self.sample(l, p)
            node_ids += order
            labels += [0.0, 0.0] + [1.0] * (len(order) - 2)
            src += [v + off for v in s]
            dst += [v + off for v in d]
            bvec += [g] * len(order)
            off += len(order)
        x = torch.cat([torch.tensor(labels, dtype=torch.float64).view(-1, 1), self.feat[torch.tensor(node_ids)]], dim=1)
        return (x.to(torch.float32), torch.tensor([src, dst], dtype=torch.long),
                torch.tensor(bvec, dtype=torch.long))


def load_checkpoint(project: str, fold: int, epoch: int) -> dict:
    path = os.path.join(REF, "result", project, f"model_{fold}_fold", str(epoch))
    return torch.load(path, map_location="cpu", weights_only=True)


def logged_metrics(project: str, fold: int, epoch: int):
    """The five %.5f strings of the 'testing dataset' line of that epoch (ASCII lines of a GBK log)."""
    text = open(os.path.join(REF, "result", project, f"log_{fold}.txt"), "rb").read().decode("latin1")
    pat = r"Accuracy: ([\d.]+), Precision: ([\d.]+), Sensitivity: ([\d.]+), Specificity: ([\d.]+), MCC: ([-\d.]+)"
    for line in text.split("\n"):
        if "testing dataset" in line and (line.startswith(f"Epoch: {epoch:03d},") or (epoch == 50 and line.startswith("result,"))):
            return list(re.search(pat, line).groups())
    raise KeyError((project, fold, epoch))


def predict(proj: Project, keys, sd, conv=R.sage_conv, chunk: int = 256) -> torch.Tensor:
    outs = []
    with torch.no_grad():
        for i in range(0, len(keys), chunk):
            ks = keys[i:i + chunk]
            x, ei, b = proj.batch(ks)
            outs.append(R.net1_forward(sd, x, ei, b, len(ks), conv=conv))
    return torch.cat(outs)


def confusion(logp: torch.Tensor, y: torch.Tensor):
    pred = logp.max(dim=1)[1]
    TP = int(((pred == 1) & (y == 1)).sum())
    FP = int(((pred == 1) & (y == 0)).sum())
    FN = int(((pred == 0) & (y == 1)).sum())
    TN = int(((pred == 0) & (y == 0)).sum())
    return TP, FN, TN, FP


def run_kat(dataset: str, project: str, fold: int, epoch: int, no_kmer: bool = False, conv=R.sage_conv,
            result_project: str = None):
    """`project` names the data directories; `result_project` (default: the same) the result/ directory
    (the noKmer runs reuse project 1223_1's keys and embeddings under result/1223_1_noKmer)."""
    proj = Project(dataset, project, fold, no_kmer)
    result_project = result_project or project
    keys = proj.test_pos + proj.test_neg
    y = torch.tensor([1] * len(proj.test_pos) + [0] * len(proj.test_neg))
    logp = predict(proj, keys, load_checkpoint(result_project, fold, epoch), conv)
    cm = confusion(logp, y)
    got = ["%.5f" % v for v in R.metrics_from_confusion(*cm)]
    return cm, got, logged_metrics(result_project, fold, epoch)


def run_kat_p(dataset: str, project: str, fold: int, epoch: int, case: str, conv=R.sage_conv):
    """max |P_oracle - P_logged| over the case study's test-fold negatives
    (written by reference src/case_study_negativeSample.py:337-355, batch_size 1)."""
    proj = Project(dataset, project, fold)
    logged = {}
    for fn in ("case_predict_positive.txt", "case_predict_negative.txt"):
        for line in open(os.path.join(REF, "data", "case_study", case, "logs", fn)):
            parts = line.rstrip("\n").split("\t")
            if len(parts) == 3:
                logged[(proj.rna_serial[parts[0]], proj.prot_serial[parts[1]])] = float(parts[2])
    keys = [k for k in proj.test_neg if k in logged]
    logp = predict(proj, keys, load_checkpoint(project, fold, epoch), conv)
    p = logp[:, 1].double().exp()
    ref = torch.tensor([logged[k] for k in keys], dtype=torch.float64)
    return float((p - ref).abs().max()), float((p - ref).abs().median()), len(keys), len(logged)


if __name__ == "__main__":
    import sys
    import time
    assert have_reference(), "needs /root/reference"
    t = time.time()
    cm, got, want = run_kat("NPInter2", "1223_1", 0, 50)
    print("KAT 1223_1 fold 0 ckpt 50:", cm, got, want, "OK" if got == want else "MISMATCH", f"{time.time()-t:.1f}s")
    if "--all" in sys.argv:
        print("RPI369:", run_kat("RPI369", "1228_1", 0, 50))
        print("noKmer:", run_kat("NPInter2", "1223_1", 0, 50, no_kmer=True, result_project="1223_1_noKmer"))
        print("KAT-P fold1:", run_kat_p("NPInter2", "1223_1", 1, 15, "1223_1_fold_1_negativeSamples_threshold_0.99"))
