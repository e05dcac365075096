"""TEST INFRASTRUCTURE -- CPU restatement of the reference's one-hop enclosing-subgraph extraction and
PyG ``Batch`` collate.  Only ``tests/`` may import this module; the product
(``npi_gnn_amd.subgraph``) never does.

Restates ``local_subgraph_generation`` (reference ``src/classes.py:652-733``) on tensor inputs:

* the interaction graph is the ordered list of (rna_serial, protein_serial) pairs -- positives in file
  order, then negatives (``interaction_list`` order, ``src/generate_dataset.py:224-305``);
* ``usable[k]`` is False for pairs in ``set_allInteractionKey_cannotUse`` (the test keys of the fold,
  ``src/generate_dataset.py:296-299``);
* a sample (l, p): local node 0 = l, 1 = p, then the protein partners of l over usable pairs in list
  order, then the RNA partners of p likewise, each numbered on first sight
  (``src/classes.py:676-692``);  edges: the target pair plus every usable pair met, each emitted in both
  directions (``:695-701``);  x = [structural label (0 for the two targets, 1 otherwise) | node
  features] (``:704-714``).

Edge ORDER: the reference iterates a Python ``set`` of pairs (``:696``), whose order is a CPython hashing
detail, not part of the algorithm (aggregation is order-free up to float rounding).  This restatement
and the device kernels emit the canonical order: target pair, pairs of l in list order, pairs of p in
list order.  Pinned in this container against ``oracle/kat.py`` (which follows the set order and
reproduces the reference's logged metrics) as equal edge SETS, identical node order, identical x.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import torch


def adjacency(pairs: Sequence[Tuple[int, int]], usable: Sequence[bool]):
    rna_adj, prot_adj = {}, {}
    for k, (l, p) in enumerate(pairs):
        rna_adj.setdefault(int(l), []).append((int(p), bool(usable[k])))
        prot_adj.setdefault(int(p), []).append((int(l), bool(usable[k])))
    return rna_adj, prot_adj


def sample(rna_adj, prot_adj, l: int, p: int):
    """-> (node ids in local order, undirected local pairs in canonical order)"""
    ids = {l: 0, p: 1}
    order = [l, p]
    und: List[Tuple[int, int]] = [(0, 1)]
    seen = {(l, p)}
    for q, ok in rna_adj.get(l, []):
        if not ok:
            continue
        if q not in ids:
            ids[q] = len(order)
            order.append(q)
        if (l, q) not in seen:
            seen.add((l, q))
            und.append((0, ids[q]))
    for m, ok in prot_adj.get(p, []):
        if not ok:
            continue
        if m not in ids:
            ids[m] = len(order)
            order.append(m)
        if (m, p) not in seen:
            seen.add((m, p))
            und.append((ids[m], 1))
    return order, und


def enclosing_subgraph_batch(pairs: torch.Tensor, usable: torch.Tensor, feat: torch.Tensor, keys: torch.Tensor):
    """pairs [P,2] int, usable [P] bool, feat [N,Ff] float, keys [B,2] int ->
    x [n, 1+Ff] fp32, edge_index [2, e] int64, batch [n] int64, node_id [n] int64"""
    rna_adj, prot_adj = adjacency(pairs.tolist(), usable.tolist())
    node_ids, labels, src, dst, bvec = [], [], [], [], []
    off = 0
    for g, (l, p) in enumerate(keys.tolist()):
        order, und = sample(rna_adj, prot_adj, l, p)
        node_ids += order
        labels += [0.0, 0.0] + [1.0] * (len(order) - 2)
        for a, b in und:
            src += [a + off, b + off]
            dst += [b + off, a + off]
        bvec += [g] * len(order)
        off += len(order)
    nid = torch.tensor(node_ids, dtype=torch.long)
    x = torch.cat([torch.tensor(labels, dtype=feat.dtype).view(-1, 1), feat[nid]], dim=1) if nid.numel() else \
        torch.zeros((0, 1 + feat.size(1)), dtype=feat.dtype)
    return (x.to(torch.float32), torch.tensor([src, dst], dtype=torch.long).view(2, -1),
            torch.tensor(bvec, dtype=torch.long), nid)
