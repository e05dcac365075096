"""Generate tests/golden/rpi7317_graph.pt (run in the build container, where /root/reference exists):
    python tests/golden/make_rpi7317.py

BASELINE.json configs[2] ("C3") on the graph SURVEY.md 8(d) specifies: RPI7317 balanced.
  nodes      serial numbers by first appearance in data/source_database_data/RPI7317.xlsx, ONE counter shared by ncRNAs
             and proteins, the RNA of a row numbered before its protein (src/generate_edgelist.py:56-88):
             1,874 ncRNAs + 118 proteins = 1,992
  positives  the 7,317 rows of the sheet (all label 1)
  negatives  7,317 pairs drawn by the rule of src/generate_edgelist.py:108-139 -- uniform ncRNA index, uniform protein
             index, redraw if the pair is a positive or already drawn -- with Python's `random` seeded 20260310 (the
             reference does not seed; no negative set of this dataset is bundled)
  edges      every pair in both directions (src/classes.py:701-704): E = 29,268
  x          [label = 1 | node2vec (absent for this dataset -> 64 zeros, src/generate_dataset.py:55-75) | 3-mer / 2-mer
             frequencies of data/lncRNA_3_mer/RPI7317, data/protein_2_mer/RPI7317], F = 178
Expected outputs are the CPU ORACLE's (3 x GCNConv 178->256->256->256, seeded glorot weights): GCNConv is never run by the
reference, parity unpinned.
"""
import os
import random
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import kat, ref_conv as R  # noqa: E402


def main():
    assert kat.have_reference(), "needs /root/reference"
    rows = kat.read_xlsx_rows(os.path.join(kat.REF, "data", "source_database_data", "RPI7317.xlsx"))[1:]
    rna, prot, pos = {}, {}, []
    n = 0
    for r in rows:
        if len(r) < 3 or r[0] == "":
            continue
        a, b, label = kat._unescape(r[0]), kat._unescape(r[1]), int(float(r[2]))
        assert label == 1
        if a not in rna:
            rna[a] = n
            n += 1
        if b not in prot:
            prot[b] = n
            n += 1
        pos.append((rna[a], prot[b]))
    assert (len(rna), len(prot), n, len(pos), len(set(pos))) == (1874, 118, 1992, 7317, 7317)
    rna_list, prot_list = list(rna.values()), list(prot.values())           # list order = first appearance
    random.seed(20260310)
    pos_set, neg_set, neg = set(pos), set(), []
    while len(neg) < len(pos):
        key = (rna_list[random.randint(0, len(rna_list) - 1)], prot_list[random.randint(0, len(prot_list) - 1)])
        if key in pos_set or key in neg_set:
            continue
        neg_set.add(key)
        neg.append(key)
    pairs = torch.tensor(pos + neg, dtype=torch.long).t()
    ei = torch.cat([pairs, pairs.flip(0)], dim=1)
    assert ei.size(1) == 29268

    kmer = torch.zeros(n, 113, dtype=torch.float64)

    def load(path, serial, lo, width):
        seen = set()
        lines = open(path).read().split("\n")
        for i, line in enumerate(lines):
            if line.startswith(">"):
                nm = line.strip()[1:]
                if nm in serial and nm not in seen:
                    seen.add(nm)
                    vals = lines[i + 1].strip().split("\t")
                    assert len(vals) == width, (nm, len(vals))
                    kmer[serial[nm], lo:lo + width] = torch.tensor([float(v) for v in vals], dtype=torch.float64)
        return len(seen)
    got_r = load(os.path.join(kat.REF, "data", "lncRNA_3_mer", "RPI7317", "lncRNA_3_mer.txt"), rna, 0, 64)
    got_p = load(os.path.join(kat.REF, "data", "protein_2_mer", "RPI7317", "protein_2_mer.txt"), prot, 64, 49)
    x = torch.cat([torch.ones(n, 1, dtype=torch.float64), torch.zeros(n, 64, dtype=torch.float64), kmer], dim=1).float()

    g = torch.Generator().manual_seed(7317)

    def glorot(i, o):
        a = (6.0 / (i + o)) ** 0.5
        return (torch.rand(i, o, generator=g) * 2 - 1) * a
    gcn256 = [(glorot(178, 256), torch.zeros(256)), (glorot(256, 256), torch.zeros(256)), (glorot(256, 256), torch.zeros(256))]
    with torch.no_grad():
        h = x
        for W, b in gcn256:
            h = torch.relu(R.gcn_conv(h, ei, W, b))
    deg = torch.bincount(ei[1], minlength=n)
    rows_keep = torch.cat([deg.argmax().view(1), torch.randperm(n, generator=g)[:399]]).unique()
    path = os.path.join(HERE, "rpi7317_graph.pt")
    torch.save({"x": x, "edge_index": ei.to(torch.int32), "rows": rows_keep, "gcn256": gcn256, "gcn256_out": h[rows_keep],
                "num_rna": len(rna), "num_protein": len(prot), "max_degree": int(deg.max()),
                "source": "RPI7317.xlsx + lncRNA_3_mer/RPI7317 + protein_2_mer/RPI7317; negatives by the rule of "
                          "src/generate_edgelist.py:108-139, random.seed(20260310); GCN weights seeded glorot"}, path)
    print("rpi7317_graph.pt", x.shape, ei.shape, "k-mer rows", got_r, got_p, "max degree", int(deg.max()),
          os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
